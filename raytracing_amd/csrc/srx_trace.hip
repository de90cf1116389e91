// raytracing_amd/csrc/srx_trace.hip — closest-hit and shadow traversal kernels (gfx950).
//
// Replaces RayEngine::iterativeBVH_AVX (Core.h:49-244), BVH_AVX::intersection (BVH.h:940-988) and
// PTri_AVX::intersection_check (Primitives.h:159-202).
//
// Mapping: one ray per OCTET (8 adjacent lanes); a wave64 carries 8 rays. Lane k of the octet is the
// reference's AVX lane k: it tests child box k of a node and triangle k of a packet, so one octet
// reads a 256-B node as two contiguous 128-B lines and a packet as three. The nearest-first child
// order (Core.h:74-106) is a rank computed with cross-lane shuffles inside the octet: keys carry the
// lane id in their low 3 bits, so they are distinct and rank == position after the reference's
// insertion sort. The traversal stack lives in LDS, one 8-byte (link, tmin) column per octet,
// interleaved so the 8 octets of a wave hit distinct banks; its depth is the exact worst case of the
// scene's tree, computed on the host (the reference reserves 256 entries, Scene.h:181).
// Bound: HBM/L2 gather bandwidth + latency (no MFMA work here); see DESIGN.md.
#include "srx_kernels.h"

#define SRX_TRACE_THREADS 256
#define SRX_OCTETS_PER_BLOCK (SRX_TRACE_THREADS / 8)


__device__ __forceinline__ uint32_t octet_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__shfl_xor((int)v, 1, 8));
    v = min(v, (uint32_t)__shfl_xor((int)v, 2, 8));
    v = min(v, (uint32_t)__shfl_xor((int)v, 4, 8));
    return v;
}

// MODE 0: closest hit.  MODE 1: shadow, API form (writes lit[]).  MODE 2: shadow, renderer queue form.
template <int MODE>
__global__ __launch_bounds__(SRX_TRACE_THREADS) void k_trace(GScene sc, TraceArgs a) {
    extern __shared__ uint2 s_stack[];   // [depth][SRX_OCTETS_PER_BLOCK]
    const int sub = threadIdx.x & 7;
    const int oct = threadIdx.x >> 3;
    const int lane = threadIdx.x & 63;
    const int oct_shift = lane & ~7;
    const uint32_t ray = blockIdx.x * SRX_OCTETS_PER_BLOCK + oct;
    const bool live = ray < a.n;

    f3 o, d;
    uint32_t pixel_sun = 0;
    if (live) {
        const float* po = a.org + (size_t)ray * a.org_stride;
        o = f3(po[0], po[1], po[2]);
        if (MODE == 2) {
            pixel_sun = a.shadow_q[ray].pixel_sun;
            const GSun& s = sc.suns[pixel_sun >> 24];
            d = f3(s.dir[0], s.dir[1], s.dir[2]);
        } else {
            const float* pd = a.dir + (size_t)ray * a.dir_stride;
            d = f3(pd[0], pd[1], pd[2]);
        }
    }
    // Core.h:53-54: fused = (-1*o)/d, inv = 1/d (IEEE divides, zeros unguarded)
    const f3 fused = (o * -1.0f) / d;
    const f3 inv = f3(1.0f) / d;

    float best_t = SRX_MISS_DISTANCE;
    float best_u = -1.0f, best_v = -1.0f;
    uint32_t best_mat = 0xFFFFFFFFu;
    int best_tri = 0;
    uint32_t n_nodes = 0, n_packets = 0;

    int sp = live ? 0 : -1;
    if (sub == 0) s_stack[oct] = make_uint2(0u, 0u);   // root: node 0, distance 0, no packets (Core.h:59-61)
    __builtin_amdgcn_wave_barrier();

    while (__any(sp >= 0)) {
        if (sp >= 0) {
            const uint2 e = s_stack[sp * SRX_OCTETS_PER_BLOCK + oct];
            sp--;
            const float dist = u2f(e.y);
            if (!(dist >= best_t)) {                                   // Core.h:65-68
                const uint32_t cnt = e.x >> SRX_LINK_COUNT_SHIFT;
                const uint32_t idx = e.x & SRX_LINK_INDEX_MASK;
                if (cnt == 0) {
                    // ---- inner node: one child box per lane (BVH.h:940-988)
                    const GNode* nd = sc.nodes + idx;
                    const float4 A = nd->a[sub];
                    const float4 B = nd->b[sub];
                    n_nodes++;
                    const float tx1 = __fmaf_rn(A.x, inv.x, fused.x), tx2 = __fmaf_rn(A.w, inv.x, fused.x);
                    const float ty1 = __fmaf_rn(A.y, inv.y, fused.y), ty2 = __fmaf_rn(B.x, inv.y, fused.y);
                    const float tz1 = __fmaf_rn(A.z, inv.z, fused.z), tz2 = __fmaf_rn(B.y, inv.z, fused.z);
                    const float tminx = mm_min(tx1, tx2), tminy = mm_min(ty1, ty2), tminz = mm_min(tz1, tz2);
                    const float tmaxx = mm_max(tx1, tx2), tmaxy = mm_max(ty1, ty2), tmaxz = mm_max(tz1, tz2);
                    const float tmin = mm_max(tminz, mm_max(tminy, tminx));
                    const float tmax = mm_min(tmaxz, mm_min(tmaxy, tmaxx));
                    const bool hit = (tmax >= 0.0f) && (tmin < tmax);
                    const uint32_t rb = hit ? f2u(tmin) : 0u;
                    // ordering key (Core.h:74-78): low 3 mantissa bits <- lane id; result==0 (either sign) is "no hit";
                    // a key that is itself +-0 is skipped by `if (num != 0)` (Core.h:86)
                    const bool allow = (rb & 0x7fffffffu) != 0u;
                    const uint32_t kb = allow ? ((rb & ~7u) | (uint32_t)sub) : 0u;
                    const bool valid = (kb & 0x7fffffffu) != 0u;
                    // total order of floats as unsigned ints (no NaN keys exist)
                    const uint32_t ord = valid ? ((kb & 0x80000000u) ? ~kb : (kb | 0x80000000u)) : 0xFFFFFFFFu;
                    int rank = 0;
#pragma unroll
                    for (int m = 1; m < 8; m++) {
                        const uint32_t other = (uint32_t)__shfl_xor((int)ord, m, 8);
                        rank += (other < ord) ? 1 : 0;
                    }
                    const unsigned long long bal = __ballot(valid);
                    const int nvalid = __popc((uint32_t)(bal >> oct_shift) & 0xFFu);
                    // push far -> near so the nearest child is popped first (Core.h:98-106); stored distance is the
                    // unmodified tmin
                    if (valid) s_stack[(sp + nvalid - rank) * SRX_OCTETS_PER_BLOCK + oct] = make_uint2(f2u(B.z), rb);
                    sp += nvalid;
                } else {
                    // ---- leaf: cnt packets of 8 triangles, one triangle per lane (Primitives.h:159-202)
                    for (uint32_t k = 0; k < cnt; k++) {
                        const GPacket* pk = sc.packets + idx + k;
                        const float4 A = pk->a[sub];
                        const float4 B = pk->b[sub];
                        const float4 C = pk->c[sub];
                        n_packets++;
                        const float e1x = A.w, e1y = B.x, e1z = B.y;
                        const float e2x = B.z, e2y = B.w, e2z = C.x;
                        // pvec = cross_avx(slope, p1p3) (VecLib.cpp:282-289)
                        const float c1 = d.z * e2y, c2 = d.x * e2z, c3 = d.y * e2x;
                        const float px = __fmaf_rn(d.y, e2z, -c1), py = __fmaf_rn(d.z, e2x, -c2), pz = __fmaf_rn(d.x, e2y, -c3);
                        const float det = __fmaf_rn(e1z, pz, __fmaf_rn(e1y, py, e1x * px));      // dot_avx :290-303
                        const float invDet = 1.0f / det;
                        const float tx = o.x - A.x, ty = o.y - A.y, tz = o.z - A.z;
                        const float q1 = tz * e1y, q2 = tx * e1z, q3 = ty * e1x;
                        const float qx = __fmaf_rn(ty, e1z, -q1), qy = __fmaf_rn(tz, e1x, -q2), qz = __fmaf_rn(tx, e1y, -q3);
                        const float uu = invDet * __fmaf_rn(tz, pz, __fmaf_rn(ty, py, tx * px));  // dot_mul_avx :309-325
                        const float vv = invDet * __fmaf_rn(d.z, qz, __fmaf_rn(d.y, qy, d.x * qx));
                        const float tt = invDet * __fmaf_rn(e2z, qz, __fmaf_rn(e2y, qy, e2x * qx));
                        const bool pass = (uu >= 0.0f) && (vv >= 0.0f) && ((uu + vv) <= 1.0f);
                        // closest-hit select (Core.h:231-239): lanes in order, strict '<' => lowest lane wins ties
                        const bool cand = pass && (tt > 0.0f) && (tt < best_t);
                        if (__ballot(cand) == 0ull) continue;
                        const uint32_t tb = cand ? f2u(tt) : 0xFFFFFFFFu;
                        const uint32_t m = octet_min_u32(tb);
                        if (m != 0xFFFFFFFFu) {
                            const unsigned long long wb = __ballot(tb == m);
                            const int w = __ffs((int)((uint32_t)(wb >> oct_shift) & 0xFFu)) - 1;
                            best_t = u2f(m);
                            best_u = __shfl(uu, w, 8);
                            best_v = __shfl(vv, w, 8);
                            best_mat = (uint32_t)__shfl((int)f2u(C.y), w, 8);
                            best_tri = __shfl((int)f2u(C.z), w, 8);
                            if (MODE != 0 && best_t <= 9999.0f) sp = -1;   // any occluder nearer than 9999 => shadowed
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (live && sub == 0) {
        if (MODE == 0) {
            GHit h;
            const bool hit = best_mat != 0xFFFFFFFFu;
            h.material = best_mat;
            h.t = hit ? best_t : -1.0f;
            h.u = best_u; h.v = best_v;
            h.tri_index = best_tri;
            a.hits[ray] = h;
            if (a.advanced_org3) {
                const f3 p = o + (best_t * d) * 0.999f;
                a.advanced_org3[3 * (size_t)ray] = p.x; a.advanced_org3[3 * (size_t)ray + 1] = p.y; a.advanced_org3[3 * (size_t)ray + 2] = p.z;
            }
        } else if (MODE == 1) {
            a.lit[ray] = best_t > 9999.0f ? 1 : 0;   // SunLight::intersects (Primitives.h:44-50)
        }
    }
    if (a.counters) {
        // one atomic per wave and counter, spread over replicated slots (a single hot word serialises the chip)
        uint32_t wn = (live && sub == 0) ? n_nodes : 0u, wp = (live && sub == 0) ? n_packets : 0u;
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) {
            wn += (uint32_t)__shfl_xor((int)wn, off, 64);
            wp += (uint32_t)__shfl_xor((int)wp, off, 64);
        }
        if (lane == 0) {
            unsigned long long* c = a.counters + (size_t)(blockIdx.x & a.counter_slot_mask) * a.counter_slot_stride;
            atomicAdd(&c[0], (unsigned long long)wn);
            atomicAdd(&c[1], (unsigned long long)wp);
        }
    }
    if (MODE == 2 && live && sub < 3 && best_t > 9999.0f) {
        const GShadow& s = a.shadow_q[ray];
        const float c = sub == 0 ? s.cr : (sub == 1 ? s.cg : s.cb);
        if (c != 0.0f) atomicAdd(&a.sample3[3 * (size_t)(pixel_sun & 0x00FFFFFFu) + sub], c);
    }
}

extern "C" hipError_t srx_launch_trace(int mode, const GScene& sc, const TraceArgs& a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    const uint32_t blocks = (a.n + SRX_OCTETS_PER_BLOCK - 1) / SRX_OCTETS_PER_BLOCK;
    const size_t lds = (size_t)sc.stack_depth * SRX_OCTETS_PER_BLOCK * sizeof(uint2);
    if (mode == 0) hipLaunchKernelGGL(k_trace<0>, dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
    else if (mode == 1) hipLaunchKernelGGL(k_trace<1>, dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
    else hipLaunchKernelGGL(k_trace<2>, dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
    return hipGetLastError();
}
