// oracle/ref_build/ref_veclib_probe.cpp — runs the reference's own VecLib.cpp object code on numbers from a file.
// Compiled against the reference's VecLib.h where it lies. Test infrastructure only: tests use it to pin the oracle's
// polynomial trig (oracle/sro_math.h full_sin / full_cos / tan) and the three AVX helpers the reference's
// packet test is made of (cross_avx, dot_avx, dot_mul_avx: VecLib.cpp:282-325) bit for bit.
// A program, not a library: VecLib.o refers to XorGen (XorRandom.cpp does not compile outside MSVC), which
// is left unresolved at link time; nothing called here reaches it.
//
// usage: ref_veclib_probe in.bin out.bin
//   in : u32 n, u32 m, n floats x, m x 8 x 10 floats (two 3-vectors of 8 lanes each: a.xyz, b.xyz, c.xyz, s) lane-major per field
//   out: n x {full_sin, full_cos, tan} floats, m records of 5 fields x 8 lanes {cross.xyz, dot, dot_mul}
#include "VecLib.h"
#include <cstdint>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    uint32_t n = 0, m = 0;
    if (std::fread(&n, 4, 1, f) != 1 || std::fread(&m, 4, 1, f) != 1) return 4;
    std::vector<float> x(n), v((size_t)m * 80);
    if (n && std::fread(x.data(), 4, n, f) != n) return 4;
    if (m && std::fread(v.data(), 4, v.size(), f) != v.size()) return 4;
    std::fclose(f);
    std::vector<float> trig((size_t)n * 3), vec((size_t)m * 8 * 5);
    for (uint32_t i = 0; i < n; i++) {
        trig[3 * i] = VecLib::full_sin(x[i]);
        trig[3 * i + 1] = VecLib::full_cos(x[i]);
        trig[3 * i + 2] = VecLib::tan(x[i]);
    }
    for (uint32_t i = 0; i < m; i++) {
        const float* p = v.data() + (size_t)i * 80;          // 10 fields x 8 lanes
        m256_vec3 a, b, c, cr;
        a.X = _mm256_loadu_ps(p); a.Y = _mm256_loadu_ps(p + 8); a.Z = _mm256_loadu_ps(p + 16);
        b.X = _mm256_loadu_ps(p + 24); b.Y = _mm256_loadu_ps(p + 32); b.Z = _mm256_loadu_ps(p + 40);
        c.X = _mm256_loadu_ps(p + 48); c.Y = _mm256_loadu_ps(p + 56); c.Z = _mm256_loadu_ps(p + 64);
        const __m256 s = _mm256_loadu_ps(p + 72);
        __m256 d, dm;
        VecLib::cross_avx(a, b, cr);                         // VecLib.cpp:282-289
        VecLib::dot_avx(a, b, d);                            // :290-303
        VecLib::dot_mul_avx(a, c, s, dm);                    // :309-325
        float* o = vec.data() + (size_t)i * 40;
        _mm256_storeu_ps(o, cr.X); _mm256_storeu_ps(o + 8, cr.Y); _mm256_storeu_ps(o + 16, cr.Z);
        _mm256_storeu_ps(o + 24, d); _mm256_storeu_ps(o + 32, dm);
    }
    f = std::fopen(argv[2], "wb");
    if (!f) return 5;
    std::fwrite(trig.data(), 4, trig.size(), f);
    std::fwrite(vec.data(), 4, vec.size(), f);
    std::fclose(f);
    return 0;
}
