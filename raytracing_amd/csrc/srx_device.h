// raytracing_amd/csrc/srx_device.h — device-side layouts and arithmetic for the trace-and-shade path.
//
// HBM layouts (all immutable after srx_scene_create):
//   geom    ONE allocation of 128-B lines: all nodes (2 lines each), then all packets (3 lines each). A link
//           is a line index into it (bits 0..24); bit 31 marks a leaf's packet, bit 30 an origin-extended leaf box and bits 25..29 count the
//           packets of the leaf that follow it. The byte offset of a lane's 16-B slot is
//           (link << 7) | (lane << 4), one VALU op (the shift drops the flag bits), and the load uses one SGPR
//           base for both kinds.
//   GNode   256 B = two 128-B lines of child boxes. For a leaf child the reference's box is the bound of the
//           leaf's triangles grown to contain the ORIGIN (quirk Q1, BVH.h:27-34), so rays enter most leaves without
//           any chance of a hit (7.4 packet tests per ray on the bench scene). Such a child stores its TIGHT planes
//           and sets SRX_LINK_EXT in its link: the reference's slab interval per axis is then
//           [min(near, z), max(far, z)] with z = fma(0, 1/d, -o/d) (exact, see k_trace), and the same six plane
//           times, padded, tell whether the ray can hit the leaf at all: if not, the leaf is not pushed (2.4 packet
//           tests per ray). Results, visiting order and node tests are the reference's.
//           Line A: 8 x {min.x, min.y, min.z, max.x}; line B: 8 x
//           {max.y, max.z, link, cone}; link == 0 => unused child slot (the root is never a child).
//           cone (leaf guard, SRX_LINK_EXT children only, else 0): bytes 0-2 = a as int8, the leaf's mean triangle
//           normal scaled to a largest component of +-127, byte 3 = h: the leaf may be skipped only by a ray with
//           |a . dq| >= 127 h (dq = unit direction x 127 as int8: one v_dot4), i.e. one that is not close to the plane
//           of any of the leaf's triangles (h = 255: never).
//           One octet (8 adjacent lanes = the reference's 8 AVX lanes, BVH.h:874-881) reads a node with
//           two dwordx4 loads per lane, each load covering one contiguous 128-B line per octet.
//   GPacket 384 B = three 128-B lines: 8 x {p1.x,p1.y,p1.z,e1.x} | 8 x {e1.y,e1.z,e2.x,0} |
//           8 x {e2.y,e2.z}, material[8], tri_index[8]; e1 = p1p2, e2 = p1p3 (Primitives.h:127-134). A test
//           reads 16 + 12 + 8 = 36 B per lane, a node test 16 + 12 (the data-return path of the vector cache is the busiest unit of the trace
//           kernel); material and tri_index are read once per ray, for the winning lane only.
//   texture RGB8 rows (Texture.h:147-159) -> RGBX8, 32x32-texel tiles of 4 KB, Z-order inside a tile: the texels
//           around a hit point share a few 128-B lines whatever the direction of the UV gradient.
//   GTri    160 B = ten 16-B rows with what shading reads of PackagedTri (Primitives.h:62-81).
// Arithmetic: compiled with -ffp-contract=off; an FMA appears only where the reference has an FMA
// intrinsic (__fmaf_rn), min/max follow x86 vminps/vmaxps operand order (second operand on NaN).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SRX_MISS_DISTANCE 99999999999999999.f   // Coredefs.cpp:9
#define SRX_LINK_LEAF 0x80000000u
#define SRX_LINK_LINE_MASK 0x01FFFFFFu       // 2^25 lines of 128 B = 4 GB of nodes + packets
#define SRX_LINK_MORE_SHIFT 25               // bits 25..29: packets of the same leaf after this one
#define SRX_LINK_MORE_MAX 31u
#define SRX_SPILL_SLOTS 16384u               // > 8 192 = the wave slots of the chip: a free area always exists
#define SRX_SPILL_MIN_LDS_DEPTH 16           // half the LDS part moves at a time and a node pushes up to 8 entries
#define SRX_LINK_EXT 0x40000000u             // leaf child whose reference box = tight planes stored here, grown to the origin
#define SRX_FILTER_PAD_REL 5.0e-4f           // leaf filter: a leaf's box is padded by this x the largest coordinate distance from the ray's origin to the box ...
#define SRX_FILTER_PAD_ABS 4.0e-6f           // ... + this x (scene size + largest |origin coordinate|), for the rounding of the slab test itself
#define SRX_CONE_NEVER 0xFF000000u           // leaf guard word: this leaf is never skipped (255 x 127 > any |a . dq|)

struct GNode { float4 a[8]; float4 b[8]; };
struct GPacket { float4 a[8]; float4 b[8]; float2 c[8]; uint32_t material[8]; uint32_t tri_index[8]; };
struct GTri {
    float4 r[10];
    // r0: UV_basis.xy, U_delta.xy      r1: V_delta.xy, material(bits), normal.x
    // r2: normal.yz, origin_offset.xy  r3: origin_offset.z, n1.xyz
    // r4: n2.xyz, n3.x                 r5: n3.yz, t1.xy
    // r6: t1.z, t2.xyz                 r7: t3.xyz, b1.x
    // r8: b1.yz, b2.xy                 r9: b2.z, b3.xyz
};
static_assert(sizeof(GNode) == 256 && sizeof(GPacket) == 384 && sizeof(GTri) == 160, "layout");

struct GMaterial {       // srx_material with texture ids resolved to table slots (-1 = constant)
    float color[3];     int color_tex;
    float roughness;    int roughness_tex;
    float metallic;     int metallic_tex;
    float specular;     int specular_tex;
    float emissive[3];  int emissive_tex;
    float normal[3];    int normal_tex;
    float transmission; int transmission_tex;
    float ior;          int ior_tex;
    int use_normals;
};
struct GTexture {        // Texture.h:69-75; texels re-tiled on upload, see tex_texel() in srx_shade.hip
    const uint32_t* data;         // RGBX8 texels: 32x32 tiles in row-major tile order, Morton order inside a tile
    int res_x, res_y;
    double U_multiplier, V_multiplier;
    int normal_transform;
    int tiles_x;                  // tiles per tile row = ceil(res_x / 32)
};
struct GSun { float dir[3]; float emission[3]; };

struct GScene {
    const char* geom;             // nodes then packets, 128-B lines (see above)
    const GTri* tris;
    const GMaterial* materials;
    const GTexture* textures;
    const GSun* suns;
    const float* pdf_lookup;      // 1024 x 1024
    const float* arccos_lookup;   // 2048
    uint32_t n_nodes, n_packets, n_tris, n_materials, n_textures, n_suns;
    int max_generations;
    int bounce_count[32];         // int(monte_carlo_max * pow(modifier, g)), host-computed (Core.h:428)
    int stack_depth;              // worst-case number of live traversal-stack entries of this tree (per ray)
    // Overflow of the traversal stack when a launch keeps fewer than stack_depth entries per octet in LDS (deep
    // trees): SRX_SPILL_SLOTS areas of 8 x stack_depth entries, one per wave that ever spills, claimed with an atomic.
    unsigned long long* spill_area;
    uint32_t* spill_ctl;          // [0] claim cursor, [1] error flag (no free area), [2] spills so far, [16 + i] area i taken
    float filter_pad;             // leaf filter: SRX_FILTER_PAD_ABS x the scene's size (the absolute part of the pad); inf: filter off
};

// ---- ray / hit / shadow records of the wavefront queues --------------------
struct GRay {            // 48 B = 3 x 16 B
    float ox, oy, oz, dx;
    float dy, dz, cr, cg;         // c = path throughput (product of coefficient/bounce_count, Core.h:594-597)
    float cb; uint32_t pixel; uint32_t key_lo, key_hi;
};
struct GHit { uint32_t material; float t, u, v; int32_t tri_index; };   // = srx_hit
struct GShadow {         // 32 B
    float ox, oy, oz; uint32_t pixel_sun;   // pixel | sun index << 24
    float cr, cg, cb; float pad;
};
static_assert(sizeof(GRay) == 48 && sizeof(GHit) == 20 && sizeof(GShadow) == 32, "records");

// ---- exact helpers ---------------------------------------------------------
__device__ __forceinline__ float mm_min(float a, float b) { return a < b ? a : b; }   // vminps
__device__ __forceinline__ float mm_max(float a, float b) { return a > b ? a : b; }   // vmaxps
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

struct f3 {
    float x, y, z;
    __device__ __forceinline__ f3() : x(0), y(0), z(0) {}
    __device__ __forceinline__ explicit f3(float s) : x(s), y(s), z(s) {}
    __device__ __forceinline__ f3(float a, float b, float c) : x(a), y(b), z(c) {}
};
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3(a.x * s, a.y * s, a.z * s); }   // XYZ.cpp:428
__device__ __forceinline__ f3 operator/(f3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 operator/(f3 a, f3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // XYZ.cpp:279
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {                                                  // XYZ.cpp:292
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float mag3(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ f3 normalize3(f3 v) {                                                    // XYZ.cpp:237
    float len = mag3(v);
    return f3(v.x / len, v.y / len, v.z / len);
}
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }   // std::min
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }   // std::max
__device__ __forceinline__ float cdot3(f3 a, f3 b) {                                                // XYZ.cpp:282
    return std_min(1.0f, std_max(0.0f, a.x * b.x + a.y * b.y + a.z * b.z));
}
__device__ __forceinline__ f3 reflect3(f3 v, f3 pole) {                                             // XYZ.cpp:178-186
    float vm = mag3(v), pm = mag3(pole);
    float mag_ratio = vm / pm;
    float cosine = dot3(v, pole) / (mag3(v) * mag3(pole));
    f3 reflection_point = pole * (cosine * mag_ratio);
    f3 pointing = reflection_point - v;
    return v + (pointing * 2.0f);
}
__device__ __forceinline__ f3 linear_mix3(float c, f3 f, f3 s) {                                    // XYZ.cpp:397
    float i = 1 - c;
    return f3(f.x * i + s.x * c, f.y * i + s.y * c, f.z * i + s.z * c);
}
__device__ __forceinline__ f3 apply_tbn(f3 p, f3 T, f3 B, f3 N) { return p.x * T + p.y * B + p.z * N; }   // Matrix.h:62
__device__ __forceinline__ f3 apply_rotation(f3 p, float qx, float qy, float qz, float qw) {       // XYZ.cpp:679-683
    f3 r(qx, qy, qz);
    f3 t = 2.0f * cross3(r, p);
    return p + qw * t + cross3(r, t);
}
__device__ __forceinline__ bool well_defined(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }   // XYZ.cpp:321

// ---- polynomial trig (VecLib.cpp:23-39,186-223,250-267) ---------------------
#define SRX_FPI 3.14159265358979323846264f
#define SRX_PI 3.14159265358979323846264
__device__ __forceinline__ float full_sin(float x) {
    float y = SRX_FPI - x;
    float y2 = y * y;
    float v = __fmaf_rn(y2, -1.926853e-08f, 2.666575e-06f);
    v = __fmaf_rn(v, y2, -0.0001977116f);
    v = __fmaf_rn(v, y2, 0.008330478f);
    v = __fmaf_rn(v, y2, -0.1666613f);
    y2 = __fmaf_rn(v, y2, 0.9999967f);
    return y * y2;
}
__device__ __forceinline__ float full_cos(float x) {
    float y = x - SRX_FPI;
    y = y * y;
    float v = __fmaf_rn(y, -1.7094536e-09f, 2.7113737e-07f);
    v = __fmaf_rn(v, y, -2.478466e-05f);
    v = __fmaf_rn(v, y, 0.001388931f);
    v = __fmaf_rn(v, y, -0.04166709f);
    v = __fmaf_rn(v, y, 0.5000007f);
    return __fmaf_rn(y, v, -1.0000001f);
}
__device__ __forceinline__ float vtan(float x) { return full_sin(x + SRX_FPI) / full_cos(x + SRX_FPI); }

// ---- keyed RNG: one splitmix64 counter per ray -------------------------------
// Float conversions are the reference's (XorRandom.cpp:37-46 pool halves, :119-121 scalars); the
// seeding is this renderer's own (the reference seeds worker threads from the clock, SceneManager.h:122).
struct KeyedGen {
    uint64_t s;
    float pool[8];
    __device__ __forceinline__ static uint64_t mix(uint64_t z) {
#ifdef SRX_EXP_CHEAPRNG   // timing experiment only (a different stream): what the 64-bit multiplies cost
        uint32_t a = (uint32_t)z, b = (uint32_t)(z >> 32);
        a = (a ^ (a >> 16)) * 0x7feb352du; b = (b ^ (b >> 15)) * 0x846ca68bu; a ^= b; b += a;
        a = (a ^ (a >> 15)) * 0x846ca68bu; b = (b ^ (b >> 16)) * 0x7feb352du;
        return ((uint64_t)(b ^ (b >> 16)) << 32) | (a ^ (a >> 16));
#endif
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ __forceinline__ uint64_t next64() { s += 0x9E3779B97F4A7C15ull; return mix(s); }
    __device__ __forceinline__ static uint64_t primary_key(uint64_t seed, uint64_t sample_index) {
        return mix(seed ^ mix(sample_index + 0x632BE59BD9B4E019ull));
    }
    __device__ __forceinline__ uint64_t child_key() const { return mix(s ^ 0xD1B54A32D192ED03ull); }
    __device__ __forceinline__ void regen_pool() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t r = next64();
            pool[2 * j] = u2f(((uint32_t)r & 0x007fffffu) | 0x3f800000u) - 1.0f;
            pool[2 * j + 1] = u2f(((uint32_t)(r >> 32) & 0x007fffffu) | 0x3f800000u) - 1.0f;
        }
    }
    __device__ __forceinline__ float fRand(float lo, float hi) {
        float f = (float)((double)(next64() >> 11) * 0x1.0p-53);
        return lo + f * (hi - lo);
    }
};

// ---- the reference's own generator (XorRandom.cpp:3-69, 119-132): xoshiro256+ scalar stream (`s0 + s3`) and the
// 4-lane pool whose output word is `s0 + s1`. One sequential stream: the renderer does not draw from it (section 3 of
// DESIGN.md); it defines the parity ray sets of fixed XorRandom seeds (srx_camera_rays_xorgen, srx_xorgen_stream).
struct XorGen {
    uint64_t s[4];
    uint64_t S[4][4];        // S[word][lane] = xe_seed_AVX[word], 64-bit lane `lane`
    float pool[8];
    __device__ __forceinline__ static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    __device__ __forceinline__ uint64_t xe_next() {                          // :19-34
        const uint64_t result = s[0] + s[3];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    __device__ __forceinline__ void seed(uint64_t n1, uint64_t n2, uint64_t n3, uint64_t n4) {   // :3-15
        s[0] = n1; s[1] = n2; s[2] = n3; s[3] = n4;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) S[i][j] = xe_next();
    }
    __device__ __forceinline__ void regen_pool() {                          // :35-69
        for (int j = 0; j < 4; j++) {
            const uint64_t r = S[0][j] + S[1][j];
            pool[2 * j] = u2f(((uint32_t)r & 0x007fffffu) | 0x3f800000u) - 1.0f;
            pool[2 * j + 1] = u2f(((uint32_t)(r >> 32) & 0x007fffffu) | 0x3f800000u) - 1.0f;
            const uint64_t t = S[1][j] << 17;
            S[2][j] ^= S[0][j]; S[3][j] ^= S[1][j]; S[1][j] ^= S[2][j]; S[0][j] ^= S[3][j];
            S[2][j] ^= t;
            S[3][j] = rotl(S[3][j], 45);
        }
    }
    __device__ __forceinline__ float fRand(float lo, float hi) {            // :119-132
        const float f = (float)((double)(xe_next() >> 11) * 0x1.0p-53);
        return lo + f * (hi - lo);
    }
};
