// oracle/sro_trace.h — TEST INFRASTRUCTURE ONLY (the parity oracle).
//
// CPU restatement of the closest-hit cast: 8-wide slab test, nearest-first
// child ordering, 8-triangle Moller-Trumbore, closest-hit select. Scalar loops
// over the 8 AVX lanes with std::fmaf exactly where the reference has an FMA
// intrinsic, x86 MINPS/MAXPS operand semantics for min/max.
// PARITY UNPINNED (no upstream tests; see sro_math.h header and DESIGN.md).
#pragma once
#include "sro_scene.h"

namespace sro {

// Coredefs.h:6-15, Coredefs.cpp:6-9
struct CastResults {
    uint32_t material = 0xFFFFFFFFu;
    float distance = 99999999999999999.f;
    XYZ TUV = XYZ(-1);
    XY texUV = XY(-1);
    int tri_index = 0;
};

struct TraceCounters {
    uint64_t nodes = 0;      // BVH_AVX::intersection calls (Core.h:70-72)
    uint64_t packets = 0;    // PTri_AVX::intersection_check calls (Core.h:194-229)
    int max_stack = 0;       // deepest stack_index + 1 reached
    uint64_t packets_unhittable = 0;   // of `packets`: tests in leaves whose padded tight box the ray misses (product's leaf filter)
    uint64_t unhittable_accepts = 0;   // closest-hit updates (Core.h:233) taken in such a leaf: must stay 0 for the filter to be exact
    uint64_t packets_guarded = 0;      // tests in leaves the box test alone would skip but the normal-cone guard keeps
    uint64_t unhittable_passes = 0;    // lanes with u, v in range and t > 0 in such a leaf, whether nearer than the best hit or not
    // How close the filter's soundness bound comes to being used, over every lane the reference's
    // test PASSES (u, v in range, t > 0) in a leaf for which guard condition (2) holds for the ray, in double from the
    // fp32 t, u, v: the distance between the triangle's point p1 + u e1 + v e2 and the ray's point o + t d, relative to
    // |o - p1| (the argument bounds it by 1.6e-4), and how far outside the leaf's tight box the ray's point lies
    // (Chebyshev), relative to D = the largest coordinate distance from the origin to the box (condition (1) pads by 5e-4 D).
    double max_r_over_T = 0.0, max_outside_over_D = 0.0;
    uint64_t margin_lanes = 0;
};

// Guard condition (2) of the leaf filter on its own: the ray is one the filter may act on at all, and it keeps clear of
// the planes of all the leaf's triangles by the leaf's cone word (same arithmetic as leaf_filter below).
static inline bool leaf_guard_holds(uint32_t cone, const XYZ& d, const XYZ& fused, const XYZ& inv) {
    auto finite = [](float x) { return (bitcast<uint32_t>(x) & 0x7f800000u) != 0x7f800000u; };
    if (!(finite(fused.X) && finite(fused.Y) && finite(fused.Z) && finite(inv.X) && finite(inv.Y) && finite(inv.Z))) return false;
    const float dd = std::fmaf(d.Z, d.Z, std::fmaf(d.Y, d.Y, d.X * d.X));
    if (!(dd >= 1e-12f && dd <= 1e12f)) return false;
    const float dmin = std::min(std::min(std::fabs(d.X), std::fabs(d.Y)), std::fabs(d.Z));
    if (!(dmin * dmin >= 1e-30f * dd)) return false;
    const float sc = 127.0f / std::sqrt(dd);
    const int qx = (int)std::nearbyint(d.X * sc), qy = (int)std::nearbyint(d.Y * sc), qz = (int)std::nearbyint(d.Z * sc);
    const int s = (int)(int8_t)(cone & 255u) * qx + (int)(int8_t)((cone >> 8) & 255u) * qy + (int)(int8_t)((cone >> 16) & 255u) * qz;
    return std::abs(s) >= (int)(cone >> 24) * 127;
}

// the product's leaf filter (node_test in raytracing_amd/csrc/srx_trace.hip), same arithmetic: a leaf is skipped when
//  (1) the ray's slab interval against the stored (tight) box, widened by pt at both ends, is empty or behind the ray;
//      pt = max_axis(max|t_plane| * |d_axis| * rho * max|1/d|) + (filter_pad + FILTER_PAD_ABS * max|o|) * max|1/d|,
//      i.e. rho x the largest coordinate distance from the origin to the box, in units of t, plus an absolute floor; AND
//  (2) the ray keeps clear of the planes of all the leaf's triangles: |a . dq| >= 127 h with the leaf's cone word (a, h)
//      and dq = round(127 d / |d|) as int8.
// Rays with a non-finite 1/d or -o/d component, |d|^2 outside [1e-12, 1e12] or a direction component below 1e-15 |d|
// are never filtered.
// Returns 1 = may hit (box), 0 = skipped, 2 = the box test alone would skip it but the guard (2) keeps it.
static inline int leaf_filter(const float* box6, uint32_t cone, const XYZ& o, const XYZ& d, const XYZ& fused, const XYZ& inv, float filter_pad, float pad_rel, float pad_abs) {
    auto finite = [](float x) { return (bitcast<uint32_t>(x) & 0x7f800000u) != 0x7f800000u; };
    if (!(finite(fused.X) && finite(fused.Y) && finite(fused.Z) && finite(inv.X) && finite(inv.Y) && finite(inv.Z))) return 1;
    const float dd = std::fmaf(d.Z, d.Z, std::fmaf(d.Y, d.Y, d.X * d.X));
    if (!(dd >= 1e-12f && dd <= 1e12f)) return 1;
    const float dmin = std::min(std::min(std::fabs(d.X), std::fabs(d.Y)), std::fabs(d.Z));
    if (!(dmin * dmin >= 1e-30f * dd)) return 1;
    const float omax = std::max(std::max(std::fabs(o.X), std::fabs(o.Y)), std::fabs(o.Z));
    const float maxinv = std::max(std::max(std::fabs(inv.X), std::fabs(inv.Y)), std::fabs(inv.Z));
    const float floor_t = std::fmaf(pad_abs, omax, filter_pad) * maxinv;
    const float c1 = pad_rel * maxinv;
    const float x1 = std::fmaf(box6[0], inv.X, fused.X), x2 = std::fmaf(box6[3], inv.X, fused.X);
    const float y1 = std::fmaf(box6[1], inv.Y, fused.Y), y2 = std::fmaf(box6[4], inv.Y, fused.Y);
    const float z1 = std::fmaf(box6[2], inv.Z, fused.Z), z2 = std::fmaf(box6[5], inv.Z, fused.Z);
    const float nx = std::min(x1, x2), ny = std::min(y1, y2), nz = std::min(z1, z2);
    const float fx = std::max(x1, x2), fy = std::max(y1, y2), fz = std::max(z1, z2);
    const float smin = std::max(nz, std::max(ny, nx));
    const float smax = std::min(fz, std::min(fy, fx));
    const float ex = std::max(-nx, fx) * std::fabs(d.X), ey = std::max(-ny, fy) * std::fabs(d.Y), ez = std::max(-nz, fz) * std::fabs(d.Z);
    const float pt = std::fmaf(c1, std::max(ez, std::max(ey, ex)), floor_t);
    if (!(smin - smax > pt + pt) && !(smax < -pt)) return 1;
    const float sc = 127.0f / std::sqrt(dd);
    const int qx = (int)std::nearbyint(d.X * sc), qy = (int)std::nearbyint(d.Y * sc), qz = (int)std::nearbyint(d.Z * sc);
    const int s = (int)(int8_t)(cone & 255u) * qx + (int)(int8_t)((cone >> 8) & 255u) * qy + (int)(int8_t)((cone >> 16) & 255u) * qz;
    return (std::abs(s) >= (int)(cone >> 24) * 127) ? 0 : 2;
}

static inline float mm_min(float a, float b) { return a < b ? a : b; }   // vminps: NaN -> b
static inline float mm_max(float a, float b) { return a > b ? a : b; }   // vmaxps: NaN -> b

// BVH.h:940-988
static inline void node_intersection(const BVH_AVX& n, const XYZ& fused, const XYZ& inv, float out[8]) {
    for (int l = 0; l < 8; l++) {
        float tx1 = std::fmaf(n.minX[l], inv.X, fused.X);
        float tx2 = std::fmaf(n.maxX[l], inv.X, fused.X);
        float ty1 = std::fmaf(n.minY[l], inv.Y, fused.Y);
        float ty2 = std::fmaf(n.maxY[l], inv.Y, fused.Y);
        float tz1 = std::fmaf(n.minZ[l], inv.Z, fused.Z);
        float tz2 = std::fmaf(n.maxZ[l], inv.Z, fused.Z);
        float tminx = mm_min(tx1, tx2), tminy = mm_min(ty1, ty2), tminz = mm_min(tz1, tz2);
        float tmaxx = mm_max(tx1, tx2), tmaxy = mm_max(ty1, ty2), tmaxz = mm_max(tz1, tz2);
        float tmin = mm_max(tminz, mm_max(tminy, tminx));
        float tmax = mm_min(tmaxz, mm_min(tmaxy, tmaxx));
        bool hit = (tmax >= 0.0f) && (tmin < tmax);   // _CMP_GE_OQ, _CMP_LT_OQ
        out[l] = hit ? tmin : 0.0f;                    // and_ps with the mask
    }
}

// One lane of the reference's AVX helpers (VecLib.cpp:282-325); tests/test_oracle_cpu.py pins them bit for bit against
// the reference's own object code (oracle/_ref/ref_veclib_probe).
static inline void cross_avx_lane(float ax, float ay, float az, float bx, float by, float bz, float& x, float& y, float& z) {
    const float c1 = az * by, c2 = ax * bz, c3 = ay * bx;            // :283-285
    x = std::fmaf(ay, bz, -c1); y = std::fmaf(az, bx, -c2); z = std::fmaf(ax, by, -c3);   // fmsub :286-288
}
static inline float dot_avx_lane(float ax, float ay, float az, float bx, float by, float bz) {   // :290-303
    return std::fmaf(az, bz, std::fmaf(ay, by, ax * bx));
}
static inline float dot_mul_avx_lane(float ax, float ay, float az, float bx, float by, float bz, float s) {   // :309-325
    return s * std::fmaf(az, bz, std::fmaf(ay, by, ax * bx));
}

// Primitives.h:159-202 with VecLib.cpp:282-325
static inline void packet_intersection(const PTri_AVX& T, const XYZ& o, const XYZ& d, float t[8], float u[8], float v[8]) {
    for (int l = 0; l < 8; l++) {
        const float e1x = T.p1p2X[l], e1y = T.p1p2Y[l], e1z = T.p1p2Z[l];
        const float e2x = T.p1p3X[l], e2y = T.p1p3Y[l], e2z = T.p1p3Z[l];
        float px, py, pz, qx, qy, qz;
        cross_avx_lane(d.X, d.Y, d.Z, e2x, e2y, e2z, px, py, pz);            // pvec = cross_avx(slope, p1p3)
        const float det = dot_avx_lane(e1x, e1y, e1z, px, py, pz);           // det = dot_avx(p1p2, pvec)
        const float invDet = 1.0f / det;
        const float tx = o.X - T.p1X[l], ty = o.Y - T.p1Y[l], tz = o.Z - T.p1Z[l];
        cross_avx_lane(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);               // qvec = cross_avx(tvec, p1p2)
        const float uu = dot_mul_avx_lane(tx, ty, tz, px, py, pz, invDet);
        const float vv = dot_mul_avx_lane(d.X, d.Y, d.Z, qx, qy, qz, invDet);
        const float tt = dot_mul_avx_lane(e2x, e2y, e2z, qx, qy, qz, invDet);
        const bool pass = (uu >= 0.0f) && (vv >= 0.0f) && ((uu + vv) <= 1.0f);
        t[l] = pass ? tt : 0.0f;
        u[l] = uu;
        v[l] = vv;
    }
}

// Core.h:49-244
static inline void iterativeBVH_AVX(const PackagedGeometry& data, CastResults& res, const XYZ& position,
                                    const XYZ& slope, TraceCounters* ctr) {
    XYZ fused = (-1 * position) / slope;          // Core.h:53
    XYZ inv = float_div_xyz(1, slope);            // Core.h:54
    const int STACK = 256;                        // Scene.h:181
    int stack[STACK];
    float distances[STACK];
    int tri_count[STACK];
    int slot_of[STACK];                           // oracle-side only: node * 8 + child the entry came from
    int stack_index = 0;
    stack[0] = 0; distances[0] = 0; tri_count[0] = 0; slot_of[0] = -1;
    int maxdepth = 1;
    while (stack_index >= 0) {
        const float dist = distances[stack_index];
        const int self_triangles = tri_count[stack_index];
        if (dist >= res.distance) { stack_index--; continue; }
        if (self_triangles == 0) {
            const int current_index = stack[stack_index];
            const BVH_AVX& current = data.avx_bvh[current_index];
            stack_index--;
            float persist[8];
            node_intersection(current, fused, inv, persist);
            if (ctr) ctr->nodes++;
            float results[8];
            for (int l = 0; l < 8; l++) {               // Core.h:74-78
                bool allow = (persist[l] != 0.0f);      // _CMP_NEQ_OQ (never NaN here)
                uint32_t b = (bitcast<uint32_t>(persist[l]) & ~7u) | (uint32_t)l;
                results[l] = allow ? bitcast<float>(b) : 0.0f;
            }
            int sorted_size = 0;
            for (int i = 0; i < 8; i++) {               // Core.h:84-97
                float num = results[i];
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
                unsigned sel = bitcast<uint32_t>(results[i]) & 7u;
                stack_index++;
                if (stack_index >= STACK) throw std::runtime_error("oracle traversal stack overflow (reference UB)");
                stack[stack_index] = (int)current.indexes[sel];
                distances[stack_index] = persist[sel];
                tri_count[stack_index] = (int)current.leaf_size[sel];
                slot_of[stack_index] = current_index * 8 + (int)sel;
            }
            if (stack_index + 1 > maxdepth) maxdepth = stack_index + 1;
        } else {
            int start_index = stack[stack_index];
            int verdict = 1;
            if (ctr && data.filter_enabled && !data.filter_boxes.empty() && slot_of[stack_index] >= 0)
                verdict = leaf_filter(&data.filter_boxes[(size_t)slot_of[stack_index] * 6], data.filter_cones[slot_of[stack_index]], position, slope,
                                      fused, inv, data.filter_pad, PackagedGeometry::FILTER_PAD_REL, PackagedGeometry::FILTER_PAD_ABS);
            const bool unhittable = verdict == 0;
            const bool measure = ctr && data.filter_enabled && !data.filter_boxes.empty() && slot_of[stack_index] >= 0 &&
                                 leaf_guard_holds(data.filter_cones[slot_of[stack_index]], slope, fused, inv);
            const float* mbox = measure ? &data.filter_boxes[(size_t)slot_of[stack_index] * 6] : nullptr;
            if (unhittable) ctr->packets_unhittable += (uint64_t)self_triangles;
            if (verdict == 2) ctr->packets_guarded += (uint64_t)self_triangles;
            stack_index--;
            for (int i = 0; i < self_triangles; i++) {
                const PTri_AVX& P = data.avx_tri_data[i + start_index];
                float t[8], u[8], v[8];
                packet_intersection(P, position, slope, t, u, v);
                if (ctr) ctr->packets++;
                for (int l = 0; l < 8 && measure; l++) {
                    if (!(t[l] > 0)) continue;
                    const double o3[3] = {position.X, position.Y, position.Z}, d3[3] = {slope.X, slope.Y, slope.Z};
                    const double p1[3] = {P.p1X[l], P.p1Y[l], P.p1Z[l]}, e1[3] = {P.p1p2X[l], P.p1p2Y[l], P.p1p2Z[l]}, e2[3] = {P.p1p3X[l], P.p1p3Y[l], P.p1p3Z[l]};
                    double r2 = 0, T2 = 0, outside = 0, D = 0;
                    for (int ax = 0; ax < 3; ax++) {
                        const double q = o3[ax] + (double)t[l] * d3[ax], tri = p1[ax] + (double)u[l] * e1[ax] + (double)v[l] * e2[ax];
                        r2 += (q - tri) * (q - tri);
                        T2 += (o3[ax] - p1[ax]) * (o3[ax] - p1[ax]);
                        outside = std::max(outside, std::max((double)mbox[ax] - q, q - (double)mbox[3 + ax]));
                        D = std::max(D, std::max(std::fabs((double)mbox[ax] - o3[ax]), std::fabs((double)mbox[3 + ax] - o3[ax])));
                    }
                    if (T2 > 0) ctr->max_r_over_T = std::max(ctr->max_r_over_T, std::sqrt(r2 / T2));
                    if (D > 0) ctr->max_outside_over_D = std::max(ctr->max_outside_over_D, outside / D);
                    ctr->margin_lanes++;
                }
                for (int l = 0; l < 8; l++) {           // Core.h:231-239
                    if (unhittable && t[l] > 0) { ctr->unhittable_passes++; if (t[l] < res.distance) ctr->unhittable_accepts++; }
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
    if (ctr && maxdepth > ctr->max_stack) ctr->max_stack = maxdepth;
}

// sro_trace_avx2.h: the same cast with 8-wide AVX2 intrinsics (bit-identical; the timed CPU baseline)
static inline void iterativeBVH_AVX2(const PackagedGeometry& data, CastResults& res, const XYZ& position, const XYZ& slope, TraceCounters* ctr);

// Core.h:323-329,346-350 — origin is advanced in place, even on a miss.
static inline CastResults execute_ray_cast(const PackagedGeometry& data, XYZ& position, const XYZ& slope,
                                           TraceCounters* ctr = nullptr) {
    CastResults r;
    if (data.simd) iterativeBVH_AVX2(data, r, position, slope, ctr);
    else iterativeBVH_AVX(data, r, position, slope, ctr);
    position += r.distance * slope * 0.999f;
    return r;
}

}  // namespace sro
