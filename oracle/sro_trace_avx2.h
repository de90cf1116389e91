// oracle/sro_trace_avx2.h — TEST INFRASTRUCTURE ONLY (the parity oracle / the timed CPU baseline).
//
// The closest-hit cast with the node and packet tests written as 8-wide AVX2 + FMA intrinsics, the way the
// reference writes them (BVH_AVX::intersection BVH.h:940-988, PTri_AVX::intersection_check Primitives.h:159-202,
// VecLib::cross_avx / dot_avx / dot_mul_avx VecLib.cpp:282-325): this is what bench.py times as `cpu_baseline`
// (SURVEY 8d asks for the CPU path as the reference runs it, not a scalar restatement). Same operations in the same
// order as the scalar loops of sro_trace.h, so results and counters are bit-identical (tests/test_oracle_cpu.py).
#pragma once
#include "sro_trace.h"
#include <immintrin.h>

namespace sro {

// BVH.h:940-988
static inline __m256 node_intersection_avx2(const BVH_AVX& n, const __m256 fx, const __m256 fy, const __m256 fz, const __m256 ix,
                                            const __m256 iy, const __m256 iz) {
    const __m256 tx1 = _mm256_fmadd_ps(_mm256_loadu_ps(n.minX), ix, fx), tx2 = _mm256_fmadd_ps(_mm256_loadu_ps(n.maxX), ix, fx);
    const __m256 ty1 = _mm256_fmadd_ps(_mm256_loadu_ps(n.minY), iy, fy), ty2 = _mm256_fmadd_ps(_mm256_loadu_ps(n.maxY), iy, fy);
    const __m256 tz1 = _mm256_fmadd_ps(_mm256_loadu_ps(n.minZ), iz, fz), tz2 = _mm256_fmadd_ps(_mm256_loadu_ps(n.maxZ), iz, fz);
    const __m256 tminx = _mm256_min_ps(tx1, tx2), tminy = _mm256_min_ps(ty1, ty2), tminz = _mm256_min_ps(tz1, tz2);
    const __m256 tmaxx = _mm256_max_ps(tx1, tx2), tmaxy = _mm256_max_ps(ty1, ty2), tmaxz = _mm256_max_ps(tz1, tz2);
    const __m256 tmin = _mm256_max_ps(tminz, _mm256_max_ps(tminy, tminx));
    const __m256 tmax = _mm256_min_ps(tmaxz, _mm256_min_ps(tmaxy, tmaxx));
    const __m256 hit = _mm256_and_ps(_mm256_cmp_ps(tmax, _mm256_setzero_ps(), _CMP_GE_OQ), _mm256_cmp_ps(tmin, tmax, _CMP_LT_OQ));
    return _mm256_and_ps(tmin, hit);
}

struct V8 { __m256 x, y, z; };
static inline V8 cross_avx(const V8& a, const V8& b) {                       // VecLib.cpp:282-289
    const __m256 c1 = _mm256_mul_ps(a.z, b.y), c2 = _mm256_mul_ps(a.x, b.z), c3 = _mm256_mul_ps(a.y, b.x);
    return V8{_mm256_fmsub_ps(a.y, b.z, c1), _mm256_fmsub_ps(a.z, b.x, c2), _mm256_fmsub_ps(a.x, b.y, c3)};
}
static inline __m256 dot_avx(const V8& a, const V8& b) {                     // :290-303
    return _mm256_fmadd_ps(a.z, b.z, _mm256_fmadd_ps(a.y, b.y, _mm256_mul_ps(a.x, b.x)));
}
static inline __m256 dot_mul_avx(const V8& a, const V8& b, const __m256 s) { return _mm256_mul_ps(s, dot_avx(a, b)); }   // :309-325

// Primitives.h:159-202
static inline void packet_intersection_avx2(const PTri_AVX& T, const V8& o, const V8& d, float t[8], float u[8], float v[8]) {
    const V8 e1{_mm256_loadu_ps(T.p1p2X), _mm256_loadu_ps(T.p1p2Y), _mm256_loadu_ps(T.p1p2Z)};
    const V8 e2{_mm256_loadu_ps(T.p1p3X), _mm256_loadu_ps(T.p1p3Y), _mm256_loadu_ps(T.p1p3Z)};
    const V8 pvec = cross_avx(d, e2);
    const __m256 det = dot_avx(e1, pvec);
    const __m256 inv = _mm256_div_ps(_mm256_set1_ps(1.0f), det);
    const V8 tvec{_mm256_sub_ps(o.x, _mm256_loadu_ps(T.p1X)), _mm256_sub_ps(o.y, _mm256_loadu_ps(T.p1Y)), _mm256_sub_ps(o.z, _mm256_loadu_ps(T.p1Z))};
    const V8 qvec = cross_avx(tvec, e1);
    const __m256 uu = dot_mul_avx(tvec, pvec, inv), vv = dot_mul_avx(d, qvec, inv), tt = dot_mul_avx(e2, qvec, inv);
    const __m256 zero = _mm256_setzero_ps();
    const __m256 pass = _mm256_and_ps(_mm256_and_ps(_mm256_cmp_ps(uu, zero, _CMP_GE_OQ), _mm256_cmp_ps(vv, zero, _CMP_GE_OQ)),
                                      _mm256_cmp_ps(_mm256_add_ps(uu, vv), _mm256_set1_ps(1.0f), _CMP_LE_OQ));
    _mm256_storeu_ps(t, _mm256_and_ps(tt, pass));
    _mm256_storeu_ps(u, uu);
    _mm256_storeu_ps(v, vv);
}

// Core.h:49-244 with the two tests above; counts node and packet tests only (no leaf-filter bookkeeping)
static inline void iterativeBVH_AVX2(const PackagedGeometry& data, CastResults& res, const XYZ& position, const XYZ& slope, TraceCounters* ctr) {
    const XYZ fused = (-1 * position) / slope;          // Core.h:53
    const XYZ inv = float_div_xyz(1, slope);            // Core.h:54
    const __m256 fx = _mm256_set1_ps(fused.X), fy = _mm256_set1_ps(fused.Y), fz = _mm256_set1_ps(fused.Z);
    const __m256 ix = _mm256_set1_ps(inv.X), iy = _mm256_set1_ps(inv.Y), iz = _mm256_set1_ps(inv.Z);
    const V8 o8{_mm256_set1_ps(position.X), _mm256_set1_ps(position.Y), _mm256_set1_ps(position.Z)};
    const V8 d8{_mm256_set1_ps(slope.X), _mm256_set1_ps(slope.Y), _mm256_set1_ps(slope.Z)};
    const int STACK = 256;                              // Scene.h:181
    int stack[STACK]; float distances[STACK]; int tri_count[STACK];
    int stack_index = 0;
    stack[0] = 0; distances[0] = 0; tri_count[0] = 0;
    int maxdepth = 1;
    uint64_t n_nodes = 0, n_packets = 0;
    while (stack_index >= 0) {
        const float dist = distances[stack_index];
        const int self_triangles = tri_count[stack_index];
        if (dist >= res.distance) { stack_index--; continue; }
        if (self_triangles == 0) {
            const BVH_AVX& current = data.avx_bvh[stack[stack_index]];
            stack_index--;
            alignas(32) float persist[8];
            const __m256 r = node_intersection_avx2(current, fx, fy, fz, ix, iy, iz);
            _mm256_store_ps(persist, r);
            n_nodes++;
            // Core.h:74-78: key = (bits & ~7) | lane where the result is non-zero
            const __m256 allow = _mm256_cmp_ps(r, _mm256_setzero_ps(), _CMP_NEQ_OQ);
            const __m256i key = _mm256_or_si256(_mm256_and_si256(_mm256_castps_si256(r), _mm256_set1_epi32(~7)), _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7));
            alignas(32) float results[8];
            _mm256_store_ps(results, _mm256_and_ps(_mm256_castsi256_ps(key), allow));
            int sorted_size = 0;
            for (int i = 0; i < 8; i++) {               // Core.h:84-97
                const float num = results[i];
                if (num != 0) {
                    int j = sorted_size - 1;
                    for (; j >= 0; j--) {
                        if (results[j] > num) results[j + 1] = results[j];
                        else break;
                    }
                    results[j + 1] = num;
                    sorted_size++;
                }
            }
            for (int i = sorted_size - 1; i >= 0; i--) { // Core.h:98-106
                const unsigned sel = bitcast<uint32_t>(results[i]) & 7u;
                stack_index++;
                if (stack_index >= STACK) throw std::runtime_error("oracle traversal stack overflow (reference UB)");
                stack[stack_index] = (int)current.indexes[sel];
                distances[stack_index] = persist[sel];
                tri_count[stack_index] = (int)current.leaf_size[sel];
            }
            if (stack_index + 1 > maxdepth) maxdepth = stack_index + 1;
        } else {
            const int start_index = stack[stack_index];
            stack_index--;
            for (int i = 0; i < self_triangles; i++) {
                const PTri_AVX& P = data.avx_tri_data[i + start_index];
                alignas(32) float t[8], u[8], v[8];
                packet_intersection_avx2(P, o8, d8, t, u, v);
                n_packets++;
                for (int l = 0; l < 8; l++) {           // Core.h:231-239
                    if (t[l] > 0 && t[l] < res.distance) {
                        res.distance = t[l];
                        res.material = P.materials[l];
                        res.TUV = XYZ(t[l], u[l], v[l]);
                        res.tri_index = P.index + l;
                    }
                }
            }
        }
    }
    if (ctr) { ctr->nodes += n_nodes; ctr->packets += n_packets; if (maxdepth > ctr->max_stack) ctr->max_stack = maxdepth; }
}

}  // namespace sro
