// oracle/sro_math.h — TEST INFRASTRUCTURE ONLY (the parity oracle).
//
// CPU restatement of the SpatialRay math layer: float3 ops, xoshiro RNG,
// polynomial trig, lookup tables. Nothing under oracle/ is shipped or linked by
// the product (raytracing_amd/); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it, as the checker.
//
// PARITY UNPINNED upstream: the reference has no tests or golden vectors for
// this path and its Core.h stack needs MSVC-only library constructs, so it is
// not built here. What IS pinned to the reference's own object
// code, bit for bit (tests/test_oracle_cpu.py): the XYZ / Quat vector math
// (XYZ.cpp -> oracle/_ref/libref_xyz.so), the polynomial trig of this file and
// the per-lane cross / dot helpers of the packet test in sro_trace.h (VecLib.cpp
// + commons.cpp -> oracle/_ref/ref_veclib_probe). Every function cites the
// reference file:line it follows (paths relative to
// /root/reference/spatialPartitioning/).
//
// Build with -ffp-contract=off: fused operations appear only where the
// reference writes an FMA intrinsic (std::fmaf below).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

namespace sro {

// commons.h:23-24
static const double PI_D = 3.14159265358979323846264;
static const float fPI = 3.14159265358979323846264f;

template <class T, class U> static inline T bitcast(const U& u) {
    static_assert(sizeof(T) == sizeof(U), "size");
    T t; std::memcpy(&t, &u, sizeof(T)); return t;
}

// ---------------------------------------------------------------- XY / XYZ
struct XY {
    float X = 0, Y = 0;
    XY() {}
    XY(float v) : X(v), Y(v) {}
    XY(float x, float y) : X(x), Y(y) {}
    XY operator+(const XY& o) const { return XY(X + o.X, Y + o.Y); }        // XYZ.cpp:21
    XY operator-(const XY& o) const { return XY(X - o.X, Y - o.Y); }        // XYZ.cpp:24
    XY operator*(float s) const { return XY(X * s, Y * s); }                // XYZ.cpp:27
    bool operator==(const XY& o) const { return X == o.X && Y == o.Y; }     // XYZ.cpp:33
};

struct XYZ {
    float X = 0, Y = 0, Z = 0;
    XYZ() {}
    XYZ(float s) : X(s), Y(s), Z(s) {}                                       // XYZ.cpp:93
    XYZ(float x, float y, float z) : X(x), Y(y), Z(z) {}
    float operator[](int n) const { return n == 0 ? X : (n == 1 ? Y : Z); } // XYZ.cpp:104
    float& operator[](int n) { return n == 0 ? X : (n == 1 ? Y : Z); }
    XYZ operator/(const XYZ& o) const { return XYZ(X / o.X, Y / o.Y, Z / o.Z); } // XYZ.cpp:339
    XYZ operator/(float d) const { return XYZ(X / d, Y / d, Z / d); }       // XYZ.cpp:342
    XYZ operator+(const XYZ& o) const { return XYZ(X + o.X, Y + o.Y, Z + o.Z); }
    XYZ operator-(const XYZ& o) const { return XYZ(X - o.X, Y - o.Y, Z - o.Z); }
    XYZ operator*(float m) const { return XYZ(X * m, Y * m, Z * m); }       // XYZ.cpp:357
    XYZ operator*(const XYZ& o) const { return XYZ(X * o.X, Y * o.Y, Z * o.Z); }
    XYZ& operator+=(const XYZ& o) { X += o.X; Y += o.Y; Z += o.Z; return *this; }
    XYZ& operator*=(const XYZ& o) { X *= o.X; Y *= o.Y; Z *= o.Z; return *this; }
    XYZ& operator*=(float s) { X *= s; Y *= s; Z *= s; return *this; }
    XYZ operator-() const { return XYZ(-X, -Y, -Z); }
    float magnitude() const { return std::sqrt(X * X + Y * Y + Z * Z); }    // XYZ.cpp:165
    bool wellDefined() const {                                               // XYZ.cpp:321-328
        auto ok = [](float x) { return (bitcast<uint32_t>(x) & 0x7f800000u) != 0x7f800000u; };
        return ok(X) && ok(Y) && ok(Z);
    }
    static XYZ min(const XYZ& a, const XYZ& b) {                             // XYZ.cpp:211
        return XYZ(a.X < b.X ? a.X : b.X, a.Y < b.Y ? a.Y : b.Y, a.Z < b.Z ? a.Z : b.Z);
    }
    static XYZ max(const XYZ& a, const XYZ& b) {                             // XYZ.cpp:218
        return XYZ(a.X > b.X ? a.X : b.X, a.Y > b.Y ? a.Y : b.Y, a.Z > b.Z ? a.Z : b.Z);
    }
    static float length(const XYZ& v) { return std::sqrt(v.X * v.X + v.Y * v.Y + v.Z * v.Z); }
    static XYZ normalize(const XYZ& v) {                                     // XYZ.cpp:237
        float len = length(v);
        return XYZ(v.X / len, v.Y / len, v.Z / len);
    }
    static float dot(const XYZ& a, const XYZ& b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; } // :279
    static float cdot(const XYZ& a, const XYZ& b) {                          // XYZ.cpp:282
        return std::min(1.0f, std::max(0.0f, a.X * b.X + a.Y * b.Y + a.Z * b.Z));
    }
    static float cosine(const XYZ& a, const XYZ& b) {                        // XYZ.cpp:285
        return dot(a, b) / (a.magnitude() * b.magnitude());
    }
    static XYZ cross(const XYZ& a, const XYZ& b) {                           // XYZ.cpp:292
        return XYZ(a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X);
    }
    static XYZ flip(const XYZ& p) { return XYZ(-p.X, -p.Y, -p.Z); }
    static XYZ reflect(const XYZ& vector, const XYZ& pole) {                 // XYZ.cpp:178-186
        float vm = vector.magnitude();
        float pm = pole.magnitude();
        float mag_ratio = vm / pm;
        XYZ reflection_point = pole * (cosine(vector, pole) * mag_ratio);
        XYZ pointing = reflection_point - vector;
        return vector + (pointing * 2);
    }
    static XYZ linear_mix(float c, const XYZ& f, const XYZ& s) {             // XYZ.cpp:397
        float i = 1 - c;
        return XYZ(f.X * i + s.X * c, f.Y * i + s.Y * c, f.Z * i + s.Z * c);
    }
    static float distance(const XYZ& p, const XYZ& o) {                      // XYZ.cpp:259
        float f1 = p.X - o.X, f2 = p.Y - o.Y, f3 = p.Z - o.Z;
        return std::sqrt(f1 * f1 + f2 * f2 + f3 * f3);
    }
    static XYZ slope(const XYZ& p, const XYZ& o) {                           // XYZ.cpp:265-275
        float d = distance(p, o);
        return (o - p) / d;
    }
};
static inline XYZ operator*(float s, const XYZ& p) { return p * s; }        // XYZ.cpp:428
// XYZ.cpp:431-433 — quirk Q4: `float - XYZ` ADDS. Kept on purpose.
static inline XYZ float_minus_xyz(float s, const XYZ& p) { return XYZ(p.X + s, p.Y + s, p.Z + s); }
static inline XYZ float_div_xyz(float s, const XYZ& p) { return XYZ(s / p.X, s / p.Y, s / p.Z); } // :434

struct Quat { float X = 0, Y = 0, Z = 0, W = 0; };
// XYZ.cpp:679-683
static inline XYZ applyRotation(const XYZ& p, const Quat& r) {
    XYZ rv(r.X, r.Y, r.Z);
    XYZ t = 2 * XYZ::cross(rv, p);
    return p + r.W * t + XYZ::cross(rv, t);
}
// Matrix.h:62-64
static inline XYZ applyTBNMatrix(const XYZ& p, const XYZ& T, const XYZ& B, const XYZ& N) {
    return p.X * T + p.Y * B + p.Z * N;
}

// ---------------------------------------------------------------- RNG
// Abstract generator so process_ray can run on (a) the reference's sequential
// per-thread xoshiro stream or (b) the per-ray keyed streams of the wavefront
// renderer (same float conversions, different seeding — see KeyedGen).
struct Rng {
    float pool[8];
    virtual void regen_pool() = 0;
    virtual float fRand(float lo, float hi) = 0;
    float pool_rand(int i) const { return pool[i]; }                         // XorRandom.cpp:16
    virtual ~Rng() {}
};

// XorRandom.h:11-29, XorRandom.cpp:3-69,119-132
struct XorGen : Rng {
    uint64_t s[4] = {0, 0, 0, 0};
    uint64_t S[4][4];   // S[word][lane] == xe_seed_AVX[word] lane
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    void seed(uint64_t n1, uint64_t n2, uint64_t n3, uint64_t n4) {          // :3-15
        s[0] = n1; s[1] = n2; s[2] = n3; s[3] = n4;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) S[i][j] = xe_next();
    }
    uint64_t xe_next() {                                                     // :19-34
        const uint64_t result = s[0] + s[3];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    void regen_pool() override {                                             // :35-69 (output is s0+s1)
        for (int j = 0; j < 4; j++) {
            const uint64_t r = S[0][j] + S[1][j];
            const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
            pool[2 * j] = bitcast<float>((lo & 0x007fffffu) | 0x3f800000u) - 1.0f;
            pool[2 * j + 1] = bitcast<float>((hi & 0x007fffffu) | 0x3f800000u) - 1.0f;
            const uint64_t t = S[1][j] << 17;
            S[2][j] ^= S[0][j]; S[3][j] ^= S[1][j]; S[1][j] ^= S[2][j]; S[0][j] ^= S[3][j];
            S[2][j] ^= t;
            S[3][j] = (S[3][j] << 45) | (S[3][j] >> 19);
        }
    }
    float xe_frand() { return (float)((double)(xe_next() >> 11) * 0x1.0p-53); } // :119-121
    float fRand(float lo, float hi) override {                               // :128-132
        float f = xe_frand();
        return lo + f * (hi - lo);
    }
};

// Keyed stream used by the wavefront renderer (raytracing_amd/csrc/srx_rng.h
// defines the same thing on the device; this is its CPU checker). One 64-bit
// splitmix64 counter per ray; float conversions are the reference's
// (XorRandom.cpp:37-46 for pool halves, :119-121 for scalars).
struct KeyedGen : Rng {
    uint64_t s = 0;
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t next64() { s += 0x9E3779B97F4A7C15ull; return mix(s); }
    static uint64_t primary_key(uint64_t seed, uint64_t sample_index) {
        return mix(seed ^ mix(sample_index + 0x632BE59BD9B4E019ull));
    }
    uint64_t child_key() const { return mix(s ^ 0xD1B54A32D192ED03ull); }
    void regen_pool() override {
        for (int j = 0; j < 4; j++) {
            const uint64_t r = next64();
            const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
            pool[2 * j] = bitcast<float>((lo & 0x007fffffu) | 0x3f800000u) - 1.0f;
            pool[2 * j + 1] = bitcast<float>((hi & 0x007fffffu) | 0x3f800000u) - 1.0f;
        }
    }
    float fRand(float lo, float hi) override {
        float f = (float)((double)(next64() >> 11) * 0x1.0p-53);
        return lo + f * (hi - lo);
    }
};

// ---------------------------------------------------------------- trig
// VecLib.cpp:23-39 coefficient tables.
static const float coefs_wc[7] = {1.7094536e-09f, -2.7113737e-07f, 2.478466e-05f, -0.001388931f,
                                  0.04166709f,    -0.5000007f,     1.0000001f};
static const float coefs_ws[6] = {-1.926853e-08f, 2.666575e-06f, -0.0001977116f,
                                  0.008330478f,   -0.1666613f,   0.9999967f};
// VecLib.cpp:186-191 + whole_sine_ss :250-258
static inline float full_sin(float x) {
    float y = fPI - x;
    float y2 = y * y;
    float v = std::fmaf(y2, coefs_ws[0], coefs_ws[1]);
    v = std::fmaf(v, y2, coefs_ws[2]);
    v = std::fmaf(v, y2, coefs_ws[3]);
    v = std::fmaf(v, y2, coefs_ws[4]);
    y2 = std::fmaf(v, y2, coefs_ws[5]);
    return y * y2;
}
// VecLib.cpp:215-220 + whole_cosine_ss :259-267 (coefficients pre-negated :59-67)
static inline float full_cos(float x) {
    float y = x - fPI;
    y = y * y;
    float v = std::fmaf(y, -coefs_wc[0], -coefs_wc[1]);
    v = std::fmaf(v, y, -coefs_wc[2]);
    v = std::fmaf(v, y, -coefs_wc[3]);
    v = std::fmaf(v, y, -coefs_wc[4]);
    v = std::fmaf(v, y, -coefs_wc[5]);
    return std::fmaf(y, v, -coefs_wc[6]);
}
// VecLib.cpp:221-223
static inline float vtan(float x) { return full_sin(x + fPI) / full_cos(x + fPI); }

// ---------------------------------------------------------------- tables
struct Tables {
    std::vector<float> arccos_lookup;   // 2048 entries, VecLib.cpp:391-398
    std::vector<float> pdf_lookup;      // 1024*1024, PDF.h:83-123
};

static inline void build_arccos(std::vector<float>& t) {
    t.resize(2048);
    for (int i = 0; i < 2048; i++) {
        double scalar2 = (double)2048 / 2;
        t[i] = (float)std::acos((double)i / scalar2 - 1);
    }
}
// VecLib.cpp:360-363
static inline float Lacos(const float* arccos_lookup, float input) {
    int index = ((int)((input + 1) / 2.0 * 2048)) & 2047;
    return arccos_lookup[index];
}

// PDF.h:30-82
namespace pdf {
static inline double NDF(double theta, double a) {
    double d = std::cos(theta);
    return (a * a) / (PI_D * std::pow((d * d) * (a * a - 1) + 1, 2));
}
static inline double get_pdf_weight(double theta, double a) { return std::cos(theta) * NDF(theta, a); }
static inline double find_weight(double a, double val, double start, double iteration_constant, int iterations) {
    const double epsilon = 0.00000000001;
    double check_theta = start;
    double point_value = 0;
    double high = check_theta;
    double low = start;
    while (point_value < val) {
        low = check_theta;
        check_theta += iteration_constant;
        high = check_theta;
        point_value = (check_theta - start) / PI_D * get_pdf_weight(check_theta, a);
        if (check_theta > PI_D / 2) { high = PI_D / 2; break; }
    }
    double low_v = (low - start) / PI_D * get_pdf_weight(low, a);
    double high_v = (high - start) / PI_D * get_pdf_weight(high, a);
    for (int i = 0; i < iterations; i++) {
        check_theta = (low + high) / 2;
        point_value = (check_theta - start) / PI_D * get_pdf_weight(check_theta, a);
        if (std::fabs(point_value - val) < epsilon) break;
        if ((point_value < val) && (low_v < high_v)) { low = check_theta; low_v = point_value; }
        else { high = check_theta; high_v = point_value; }
    }
    return check_theta;
}
// PDF.h:83-123
static inline void prep(std::vector<float>& lookup) {
    const int R = 1024;
    lookup.resize((size_t)R * R);
    const double segment_scalar = 1.0 / (double)R;
    const double integral_scalar = 1.0 / 1024.0;
    const double max_integration_step = integral_scalar * 2;
    const double min_integration_step = integral_scalar / 10;
    const double a_start = 0.001, a_range = 1 - a_start;
    for (int i_a = 0; i_a < R; i_a++) {
        double a = i_a * segment_scalar * a_range + a_start;
        size_t off = (size_t)i_a * R;
        double int_total = 0;
        double theta_int = -PI_D / 2;
        while (theta_int < PI_D / 2) {
            double int_step = std::pow(std::sin(theta_int), 2) * max_integration_step + min_integration_step;
            theta_int += PI_D * int_step;
            double weight = get_pdf_weight(theta_int, a);
            int_total += weight * int_step;
        }
        double v_step = int_total / (double)R;
        double t_pos = -PI_D / 2;
        double iteration = std::min(std::pow(a, 1), 0.1);
        for (int i = 0; i < R; i++) {
            t_pos = find_weight(a, v_step, t_pos, std::sin(t_pos) * std::sin(t_pos) * iteration + iteration / 100, 5);
            lookup[off + i] = (float)t_pos;
        }
    }
}
// PDF.h:124-129
static inline float sample(const float* lookup, float factor, float a) {
    int offset = (int)(std::round(a * (1024 - 1)) * 1024);
    int index = (int)std::round(factor * (1024 - 1));
    index = std::max(0, std::min(1023, index));
    return lookup[offset + index];
}
}  // namespace pdf

}  // namespace sro
