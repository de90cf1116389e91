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
// scene's tree, computed on the host (the reference reserves 256 entries, Scene.h:181), or, for trees
// too deep for that, 24 entries backed by an overflow area in memory (template parameter SPILL).
// Bound: HBM/L2 gather bandwidth + latency (no MFMA work here); see DESIGN.md.
#include "srx_kernels.h"
#include <cstdlib>
#include <vector>
#include <string>
#include <type_traits>
#include <algorithm>

// One wave per block: a finished wave frees its slot at once instead of waiting for three block mates whose rays
// run longer (measured: 256 -> 64 threads per block = -5 % kernel time on incoherent rays).
#ifndef SRX_TRACE_THREADS
#define SRX_TRACE_THREADS 64
#endif
#define SRX_OCTETS_PER_BLOCK (SRX_TRACE_THREADS / 8)
// Rays per octet: a wave carries 8 K rays and its octets work through them one after the other, so it ends when the
// pool is used up rather than with its longest single ray (K = 4 saves 7 % of the wave-trips with a fixed four rays per
// octet; an octet that ends a ray takes the next ray nobody has taken, not the next of its own four: another 2 %). All
// 8 K set-ups (the IEEE divides of Core.h:53-54) are computed up front, one ray per lane, and parked in LDS; switching
// rays costs three LDS reads.
// Measured (bench workload, before K became a template parameter): K = 1 / 2 / 3 / 4 / 5 / 6 / 8 -> 1002 / 1007 / 1031 / 1041 / 1044 / 944 / 950 Mrays/s (the
// shadow kernel gains 12 %, the closest-hit kernel nothing; from K = 6 the LDS footprint costs a wave per SIMD).
// K is a template parameter: the launcher takes the largest of 8 / 4 / 2 / 1 whose LDS footprint (64 B x LDS stack depth +
// 128 + 416 K bytes per wave) still lets 32 waves share a CU's 160 KB, because near 8 waves per SIMD the kernel's time
// goes 1:1 with the waves in flight. Results do not depend on K.
#define SRX_RAYS_PER_OCTET_MAX 8
#define SRX_LDS_PER_WAVE_FOR_FULL_OCCUPANCY 5120u
// SPILL form: 24 stack entries per octet in LDS, the rest in the overflow area, and the LDS this frees goes to a pool of
// 8 rays per octet (5 024 B per wave). A tree whose worst-case stack exceeds 25 entries cannot have that pool with its
// whole stack in LDS, one beyond 52 not even 4 rays per octet. On the bench scene (bound 47, no ray ever holds more than
// 18 entries) the plain form with 4 rays per octet runs 1403 Mrays/s on one stream, the SPILL form with 8 runs 1424 and
// with 4 only 1385: the overflow check costs about 1 % per trip, the larger pool returns 3 %. The launcher takes the
// form that carries more rays per octet for the launch at hand.
// Where it pays is measured, not derived: +2.3 % on the atrium and +1 % on the interior (bound 47), +8 % on the 10
// M-triangle atrium (bound 62, where the plain form is down to 2 rays per octet), but -4 % on the closed diffuse box
// (bound 32: every ray is long, the pool has little to even out and the check is pure cost), and launches under 4 x
// 8 192 waves of 64 rays lose more to the longer waves' tail than they gain. Hence: trees with a bound of at least 40,
// launches of at least 2.1 M rays (SRX_K8_ROUNDS, swept 1-16 on both scenes).
#define SRX_SPILL_LDS_DEPTH 20
#define SRX_SPILL_MIN_BOUND 40


// ---- cross-lane helpers inside an octet, all DPP (no LDS crossbar): quad_perm for lane^1, ^2, ^3 and
// row_half_mirror (lane -> 7-lane inside each 8-lane half row) for the other quad.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141  // row_half_mirror

__device__ __forceinline__ uint32_t octet_min_u32(uint32_t v) {
    v = min(v, dpp_u32<DPP_XOR1>(v));
    v = min(v, dpp_u32<DPP_XOR2>(v));
    v = min(v, dpp_u32<DPP_HALF_MIRROR>(v));
    return v;
}
// Number of the 7 other lanes of the octet whose key is smaller than this lane's. The subtract takes its first
// operand through DPP and leaves "other < mine" in the borrow, which the add-with-carry folds into the count:
// 2 VALU ops per neighbour (the compiler's form is mov_dpp + cmp + addc). The s_nop pairs are the gfx950 wait
// states between a VALU write of VCC / of a DPP source and its VALU read; the assembler is not told about them
// inside an asm block.
__device__ __forceinline__ int octet_rank_u32(uint32_t ord) {
    uint32_t rank = 0, h, t;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %1, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_sub_co_u32_dpp %2, vcc, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_sub_co_u32_dpp %2, vcc, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_sub_co_u32_dpp %2, vcc, %3, %3 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_lt_u32_e32 vcc, %1, %3\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_sub_co_u32_dpp %2, vcc, %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_sub_co_u32_dpp %2, vcc, %1, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_sub_co_u32_dpp %2, vcc, %1, %3 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        : "+v"(rank), "=&v"(h), "=&v"(t) : "v"(ord) : "vcc");
    return (int)rank;
}

// x86 vminps/vmaxps return the SECOND operand when either is NaN. FAST (no NaN can occur: every ray of the wave
// has finite 1/d and -o/d, so the slab FMAs give finite or infinite values only) uses the hardware min/max
// (v_min3/v_max3), which agree except for the sign of a zero result; a zero result is "no hit" either way.
// The hardware forms are written as asm: through fminf/fmaxf the compiler adds a canonicalising v_max x,x per input.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max_neg(float a, float b) { float r; asm("v_max_f32 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // max(-a, b)
__device__ __forceinline__ float x86_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float x86_max(float a, float b) { return a > b ? a : b; }

// One child box per lane (BVH.h:940-988). Returns the raw result bits (tmin or 0) of the REFERENCE's box and, in
// may_hit, whether the ray can hit anything inside the stored (tight) box at all.
//   Stored planes (srx_device.h): the reference's, except for a leaf child with SRX_LINK_EXT, whose reference box is
// the stored box grown to the origin. For such a child the reference's plane on each side of an axis is either the
// stored one or 0, and 0 lies between the stored planes or beyond them on the side the stored plane was clamped from;
// fma(x, 1/d, -o/d) is monotone in x, so the reference's slab interval is [min(near, z), max(far, z)] with
// z = fma(0, 1/d, -o/d), which is -o/d itself up to the sign of a zero (a zero result is "no hit" either way).
//   Filter: a leaf is skipped (may_hit false) when BOTH hold: (1) the stored box's interval, widened by pt on both
// ends, is empty or behind the ray, with pt = SRX_FILTER_PAD_REL x the largest coordinate distance from the ray's origin
// to the box (max|t_plane| |d_axis|, in units of t: c1 = SRX_FILTER_PAD_REL max|1/d|) + floor_t, the absolute part
// (SRX_FILTER_PAD_ABS x (scene size + max|o|) max|1/d|; infinite for rays that are never filtered); (2) the ray keeps clear of the planes of
// all the leaf's triangles, |a . dq| >= 127 h with the leaf's cone word (srx_device.h): near such a plane the
// reference's triangle test divides by a det that is rounding noise and can accept far outside the padded box
// (tests/test_leaf_filter_*.py). An inner child that the reference test accepts always passes (1)
// and its cone word is 0 (a = 0, h = 0: (2) holds trivially).
template <bool FAST, bool SDQ>
__device__ __forceinline__ uint32_t node_test(const float4 A, const float4 B, const f3 inv, const f3 fused, const f3 d, const float c1, const int dq,
                                              const float floor_t, const float zero_key_limit, bool& valid, unsigned long long& valid_mask) {
    const float tx1 = __fmaf_rn(A.x, inv.x, fused.x), tx2 = __fmaf_rn(A.w, inv.x, fused.x);
    const float ty1 = __fmaf_rn(A.y, inv.y, fused.y), ty2 = __fmaf_rn(B.x, inv.y, fused.y);
    const float tz1 = __fmaf_rn(A.z, inv.z, fused.z), tz2 = __fmaf_rn(B.y, inv.z, fused.z);
    const bool ext = f2u(B.z) >= (SRX_LINK_LEAF | SRX_LINK_EXT);   // only leaf children carry SRX_LINK_EXT (srx_scene.hip): one unsigned compare
    const float nx = hw_min(tx1, tx2), ny = hw_min(ty1, ty2), nz = hw_min(tz1, tz2);
    const float fx = hw_max(tx1, tx2), fy = hw_max(ty1, ty2), fz = hw_max(tz1, tz2);
    const float smin = hw_max3(nz, ny, nx), smax = hw_min3(fz, fy, fx);
    // guard: int8 dot of the leaf's axis with the ray's quantised direction (byte 3 of dq is 0, so h drops out of it)
#if defined(SRX_EXP_B3) || defined(SRX_EXP_NOGUARD)
    const bool clear_of_planes = true;   // experiment builds only (unsound): what the guard costs
#else
    const uint32_t cone = f2u(B.w);
    // v_dot4_i32_i8 with the constant 0 as its accumulator (the builtin becomes v_dot4c, which accumulates into its destination
    // and needs a v_mov of 0 first)
    int sd;
    if (SDQ) asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(sd) : "v"(cone), "s"(dq));
    else asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(sd) : "v"(cone), "v"(dq));
    const bool clear_of_planes = (uint32_t)abs(sd) >= (cone >> 24) * 127u;
#endif
    // largest coordinate distance from the origin to the box (the |d| are operand modifiers), then c1 x that + the floor in one FMA
    const float ex = hw_max_neg(nx, fx) * fabsf(d.x), ey = hw_max_neg(ny, fy) * fabsf(d.y), ez = hw_max_neg(nz, fz) * fabsf(d.z);
    const float pt = __fmaf_rn(c1, hw_max3(ez, ey, ex), floor_t);
    const bool not_empty = !(smin - smax > pt + pt), not_behind = !(smax < -pt);
    const bool may_hit = (not_empty && not_behind) || !clear_of_planes;
    float tmin, tmax;
    if (FAST) {
        const float emin = hw_max3(hw_min(nz, fused.z), hw_min(ny, fused.y), hw_min(nx, fused.x));
        const float emax = hw_min3(hw_max(fz, fused.z), hw_max(fy, fused.y), hw_max(fx, fused.x));
        tmin = ext ? emin : smin;
        tmax = ext ? emax : smax;
    } else {
        // a ray of the wave has a non-finite 1/d or -o/d: the reference's planes and x86 min/max, literally
        float rx1 = A.x, ry1 = A.y, rz1 = A.z, rx2 = A.w, ry2 = B.x, rz2 = B.y;
        if (ext) {
            rx1 = rx1 < 0.0f ? rx1 : 0.0f; ry1 = ry1 < 0.0f ? ry1 : 0.0f; rz1 = rz1 < 0.0f ? rz1 : 0.0f;
            rx2 = rx2 > 0.0f ? rx2 : 0.0f; ry2 = ry2 > 0.0f ? ry2 : 0.0f; rz2 = rz2 > 0.0f ? rz2 : 0.0f;
        }
        const float ux1 = __fmaf_rn(rx1, inv.x, fused.x), ux2 = __fmaf_rn(rx2, inv.x, fused.x);
        const float uy1 = __fmaf_rn(ry1, inv.y, fused.y), uy2 = __fmaf_rn(ry2, inv.y, fused.y);
        const float uz1 = __fmaf_rn(rz1, inv.z, fused.z), uz2 = __fmaf_rn(rz2, inv.z, fused.z);
        const float tminx = x86_min(ux1, ux2), tminy = x86_min(uy1, uy2), tminz = x86_min(uz1, uz2);
        const float tmaxx = x86_max(ux1, ux2), tmaxy = x86_max(uy1, uy2), tmaxz = x86_max(uz1, uz2);
        tmin = x86_max(tminz, x86_max(tminy, tminx));
        tmax = x86_min(tmaxz, x86_min(tmaxy, tmaxx));
    }
    // The reference's result is hit ? tmin : 0, and a zero result (either sign) is "no hit"; the ordering key carries the
    // lane id in its low 3 mantissa bits and a key that is itself +-0 is skipped by `if (num != 0)` (Core.h:74-78, :86): both
    // together are |tmin| > (lane == 0 ? the float with bits 7 : 0), a float compare (denormals are on; tmin is no NaN when
    // tmin < tmax holds). A leaf the ray cannot hit is not pushed either. The wave's mask of valid lanes is put together
    // from the masks of the single compares (scalar ANDs) instead of a ballot of the combined flag, which the compiler
    // turns into a select and a second compare.
    const bool in_front = tmax >= 0.0f, ordered = tmin < tmax, nonzero = fabsf(tmin) > zero_key_limit;
    valid = in_front && ordered && nonzero && may_hit;
    valid_mask = __builtin_amdgcn_ballot_w64(in_front) & __builtin_amdgcn_ballot_w64(ordered) & __builtin_amdgcn_ballot_w64(nonzero) &
                 ((__builtin_amdgcn_ballot_w64(not_empty) & __builtin_amdgcn_ballot_w64(not_behind)) | __builtin_amdgcn_ballot_w64(!clear_of_planes));
    return f2u(tmin);   // the reference's result where `valid` holds; not used elsewhere
}

// MODE 0: closest hit.  MODE 1: shadow, API form (writes lit[]).  MODE 2: shadow, renderer queue form.
// MODE 3: MODE 2 for a scene with ONE sun: every ray of the launch has the same direction (SunLight::vectorTo returns the
// stored vector whatever the point, Primitives.h:39-51), so 1/d, c1, the guard's quantised direction and d itself are
// per-launch constants held in scalar registers instead of per-ray values in LDS and vector registers.
// SPILL: the launch keeps a.lds_depth < sc.stack_depth entries per octet in LDS; the bottom half of an octet's LDS part
// moves to the wave's overflow area in memory when a push might not fit, and comes back when the LDS part runs empty.
template <int MODE, int K, bool SPILL>
__global__ __launch_bounds__(SRX_TRACE_THREADS) __attribute__((amdgpu_waves_per_eu(8))) void k_trace(GScene sc, TraceArgs a) {
    // LDS: traversal stack [depth][octets of the block] of (tmin bits << 32 | link), then one 16-B best-hit record per octet
    extern __shared__ unsigned long long s_mem[];
    // one unused row below the stack, so that the entry under an octet's only entry can be read without a bounds test
    unsigned long long* const s_stack = s_mem + SRX_OCTETS_PER_BLOCK;
    const int D = a.lds_depth;
    uint4* const s_best = (uint4*)(s_stack + (size_t)D * SRX_OCTETS_PER_BLOCK);
    const int sub = threadIdx.x & 7;
    const int oct = threadIdx.x >> 3;
    const int lane = threadIdx.x & 63;
    const int oct_shift = lane & ~7;
    const uint32_t sub16 = (uint32_t)sub << 4;
    const uint32_t sub8 = 256u + 8u * (uint32_t)sub;     // a lane's 8-B slot of a packet's third line
    const float zero_key_limit = u2f(sub == 0 ? 7u : 0u);
    // XCD-aware block -> ray-range map: blocks are dealt round-robin over the 8 XCDs (b and b+8 share one), so
    // XCD x works on the x-th contiguous eighth of the queue (queue order follows pixel / parent order).
    // Speed only; any placement gives the same results.
    // (the grid may have been sized for more rays than there are, a.n_ptr: the map is made over the blocks that have
    // rays, so that they are still dealt evenly over the XCDs; the others leave at once)
    const uint32_t n_rays = a.n_ptr ? min(*a.n_ptr, a.n) : a.n;
    // the dealing of the rays to the blocks (srx_trace_layout: full blocks, then a tail of ever smaller ones), made over the
    // rays there are; the grid was sized for a.n, blocks beyond the dealing leave at once
    const TraceLayout layout = srx_trace_layout<K>(n_rays, a.tail_blocks);
    const uint32_t nblk_used = layout.blocks;
    const uint32_t nblk = min(gridDim.x, (nblk_used + 7u) & ~7u);
    if (blockIdx.x >= nblk) return;
    // XCD x takes chunks x, x + 8, x + 16, ... of `xcd_chunk` consecutive blocks (consecutive rays: neighbours in the image or
    // children of neighbours, which walk the same part of the tree and share that XCD's L2), so all eight XCDs advance
    // through the queue together. One contiguous eighth per XCD (the first form of this map) left the XCDs with unequal
    // work: image regions differ in cost and the queue is in image order.
    const uint32_t lblock = srx_xcd_logical_block(blockIdx.x, nblk, a.xcd_chunk);
    // this block's rays: [block_start, block_start + block_rays), contiguous in the queue
    uint32_t block_start = 0;
    int block_rays = 0;
    {
        uint32_t cnt;
        srx_trace_block_range<K>(layout, n_rays, lblock, block_start, cnt);
        block_rays = (int)cnt;
    }
    constexpr bool SHQ = MODE >= 2;     // shadow rays of the renderer's queue
    constexpr bool UNI = MODE == 3;     // ... that all share one direction
    float4* const s_setup = (float4*)(s_best + SRX_OCTETS_PER_BLOCK);   // [K][octets][3]: {d, o.x}, {o.yz, inv.xy}, {inv.z, fused}
    float2* const s_pt = (float2*)(s_setup + (size_t)K * SRX_OCTETS_PER_BLOCK * 3);   // [K][octets]: leaf filter widening, quantised direction (bits)
    uint32_t* const s_spilled = (uint32_t*)(s_pt + K * SRX_OCTETS_PER_BLOCK);   // [octets]: entries parked in memory (SPILL)
    // (LDS per wave = 64 B x stack depth + 1 792 B; 5 120 B is what 32 waves per CU can have)

    // ---- set-up of all K rays of the octet at once: lane `sub` prepares ray `sub`
    bool fast;
    {
        const int myj = sub * SRX_OCTETS_PER_BLOCK + oct;       // lane `sub` of octet `oct` prepares the block's ray myj
        const uint32_t myray = block_start + (uint32_t)myj;
        const bool mine = sub < K && myj < block_rays;
        f3 so, sd;
        if (mine) {
            const float* po = a.org + (size_t)myray * a.org_stride;
            so = f3(po[0], po[1], po[2]);
            if (UNI) {
                sd = f3(sc.suns[0].dir[0], sc.suns[0].dir[1], sc.suns[0].dir[2]);
            } else if (SHQ) {
                const GSun& s = sc.suns[a.shadow_q[myray].pixel_sun >> 24];
                sd = f3(s.dir[0], s.dir[1], s.dir[2]);
            } else {
                const float* pd = a.dir + (size_t)myray * a.dir_stride;
                sd = f3(pd[0], pd[1], pd[2]);
            }
        }
        // Core.h:53-54: fused = (-1*o)/d, inv = 1/d (IEEE divides, zeros unguarded)
        const f3 sfused = (so * -1.0f) / sd;
        const f3 sinv = f3(1.0f) / sd;
        const bool finite6 = well_defined(sfused.x) && well_defined(sfused.y) && well_defined(sfused.z) &&
                             well_defined(sinv.x) && well_defined(sinv.y) && well_defined(sinv.z);
        fast = !__any(mine && !finite6);   // wave-uniform
        // the absolute part of the leaf filter's widening of this ray's slab interval (node_test): (filter_pad +
        // SRX_FILTER_PAD_ABS * max|o|) * max|1/d|; infinite = never filtered, for a ray with a non-finite 1/d or -o/d
        // component, a direction whose squared length leaves [1e-12, 1e12] or one with a component below 1e-15 |d| (the
        // guard and the relative part of the widening need them in range)
        const float sdd = __fmaf_rn(sd.z, sd.z, __fmaf_rn(sd.y, sd.y, sd.x * sd.x));
        const float sdmin = fminf(fminf(fabsf(sd.x), fabsf(sd.y)), fabsf(sd.z));
        const float spad = __fmaf_rn(SRX_FILTER_PAD_ABS, fmaxf(fmaxf(fabsf(so.x), fabsf(so.y)), fabsf(so.z)), sc.filter_pad);
        const float spt = (finite6 && sdd >= 1e-12f && sdd <= 1e12f && sdmin * sdmin >= 1e-30f * sdd)
                              ? spad * fmaxf(fmaxf(fabsf(sinv.x), fabsf(sinv.y)), fabsf(sinv.z)) : __builtin_inff();
        // the guard's quantised direction: round(127 d / |d|) as three int8 (anything when spt is infinite: never used)
        const float sq = 127.0f / sqrtf(sdd);
        const uint32_t sdq = ((uint32_t)(int)rintf(sd.x * sq) & 255u) | (((uint32_t)(int)rintf(sd.y * sq) & 255u) << 8) | (((uint32_t)(int)rintf(sd.z * sq) & 255u) << 16);
        if (sub < K) {
            float4* dst = s_setup + ((size_t)sub * SRX_OCTETS_PER_BLOCK + oct) * 3;
            dst[0] = make_float4(sd.x, sd.y, sd.z, so.x);
            dst[1] = make_float4(so.y, so.z, sinv.x, sinv.y);
            dst[2] = make_float4(sinv.z, sfused.x, sfused.y, sfused.z);
            s_pt[sub * SRX_OCTETS_PER_BLOCK + oct] = make_float2(spt, u2f(sdq));
        }
    }
    // the block's rays that exist are j = 0 .. block_rays-1 (ray j was set up by lane j / 8 of octet j % 8); octet o starts
    // with ray o, `next_j` (wave-uniform) is the first ray no octet has taken yet
    int cur = oct < block_rays ? oct : -1;   // the ray this octet works on, -1 = none
    int next_j = SRX_OCTETS_PER_BLOCK;
    uint32_t ray = block_start + (uint32_t)oct;
    const bool live = cur >= 0;
    __builtin_amdgcn_wave_barrier();

    // per ray in registers: what the node test needs (1/d, -o/d); origin and direction stay in LDS and are read by the
    // few trips that need them (packet tests, results): fewer loop-carried registers
    f3 inv, fused;
    const float4* ray_setup = s_setup;
    float c1 = 0.0f, floor_t = 0.0f;
    int dq = 0;
    // UNI: the launch's one direction and what derives from it, in scalar registers (the same arithmetic as the per-ray
    // set-up above, on the same numbers: d comes from the same GSun)
    f3 u_d;
    if (UNI) {
        const f3 ud(sc.suns[0].dir[0], sc.suns[0].dir[1], sc.suns[0].dir[2]);
        const f3 ui = f3(1.0f) / ud;
        const float uc1 = SRX_FILTER_PAD_REL * fmaxf(fmaxf(fabsf(ui.x), fabsf(ui.y)), fabsf(ui.z));
        const float udd = __fmaf_rn(ud.z, ud.z, __fmaf_rn(ud.y, ud.y, ud.x * ud.x));
        const float uq = 127.0f / sqrtf(udd);
        const uint32_t udq = ((uint32_t)(int)rintf(ud.x * uq) & 255u) | (((uint32_t)(int)rintf(ud.y * uq) & 255u) << 8) | (((uint32_t)(int)rintf(ud.z * uq) & 255u) << 16);
        auto sgpr = [](float x) { return u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(x))); };
        u_d = f3(sgpr(ud.x), sgpr(ud.y), sgpr(ud.z));
        inv = f3(sgpr(ui.x), sgpr(ui.y), sgpr(ui.z));
        c1 = sgpr(uc1);
        dq = __builtin_amdgcn_readfirstlane((int)udq);
    }
    auto load_ray_j = [&](int j) {
        ray_setup = s_setup + (size_t)j * 3;
        const float4 r2 = ray_setup[2];
        const float2 pq = s_pt[j];
        floor_t = pq.x; fused = f3(r2.y, r2.z, r2.w);
        if (!UNI) {
            const float2 r1 = *(const float2*)((const char*)ray_setup + 24);
            dq = (int)f2u(pq.y); inv = f3(r1.x, r1.y, r2.x);
            c1 = SRX_FILTER_PAD_REL * fmaxf(fmaxf(fabsf(inv.x), fabsf(inv.y)), fabsf(inv.z));
        }
    };
    load_ray_j(oct);

    // closest hit: the nearest hit so far. The shadow forms keep no distance: a ray ends at its first occluder within 9999
    // (sp = -2 tells finish() it was occluded), a hit beyond 9999 changes nothing (Primitives.h:44-50), and without a distance
    // there is no cull at the pop: it could only skip boxes beyond a hit that is itself beyond 9999, in scenes larger than that.
    float best_t = SRX_MISS_DISTANCE;
    int sp = live ? 0 : -1;              // top of the octet's stack; -1: empty, -2 (shadow forms): the ray ended at an occluder
    uint32_t n_nodes = 0, n_packets = 0;

    // what the reference returns for one ray (CastResults, Core.h:231-243; SunLight::intersects, Primitives.h:44-50)
    auto finish = [&](uint32_t r) {
        if (sub == 0) {
            if (MODE == 0) {
                GHit h;
                const bool hit = best_t < SRX_MISS_DISTANCE;
                const uint4 b = s_best[oct];
                h.material = 0xFFFFFFFFu; h.tri_index = 0;
                if (hit) {   // third line of the winning packet: ... material[8] at +64, tri_index[8] at +96
                    h.material = *(const uint32_t*)(sc.geom + (size_t)b.z);
                    h.tri_index = *(const int32_t*)(sc.geom + (size_t)b.z + 32);
                }
                h.t = hit ? best_t : -1.0f;
                h.u = hit ? u2f(b.x) : -1.0f; h.v = hit ? u2f(b.y) : -1.0f;
                a.hits[r] = h;
                if (a.advanced_org3) {
                    const float4 r0 = ray_setup[0];
                    const float2 r1 = *(const float2*)(ray_setup + 1);
                    const f3 o(r0.w, r1.x, r1.y), d(r0.x, r0.y, r0.z);
                    const f3 p = o + (best_t * d) * 0.999f;
                    a.advanced_org3[3 * (size_t)r] = p.x; a.advanced_org3[3 * (size_t)r + 1] = p.y; a.advanced_org3[3 * (size_t)r + 2] = p.z;
                }
            } else if (MODE == 1) {
                a.lit[r] = sp != -2 ? 1 : 0;
            }
        }
        if (SHQ && sub < 3 && sp != -2) {
            const GShadow& s = a.shadow_q[r];
            const float c = sub == 0 ? s.cr : (sub == 1 ? s.cg : s.cb);
            if (c != 0.0f) atomicAdd(&a.sample3[3 * (size_t)(s.pixel_sun & 0x00FFFFFFu) + sub], c);
        }
    };

    if (sub == 0) s_stack[oct] = 0ull;   // root: node 0, distance 0, no packets (Core.h:59-61)
    if (SPILL && sub == 0) s_spilled[oct] = 0u;
    int spill_slot = -1;                 // wave-uniform: this wave's overflow area, claimed at its first spill
    unsigned long long spilled_mask = 0ull;   // wave-uniform: lanes of the octets that have entries in memory
    __builtin_amdgcn_wave_barrier();

    while (__builtin_amdgcn_ballot_w64(sp >= 0) != 0ull) {
        // A wave that is about to pop and request its next lines goes first, one that has its data and computes goes
        // last: the sooner a trip's loads are out, the more of their latency the other waves cover (+0.4 %; the
        // opposite order costs 4 %).
        __builtin_amdgcn_s_setprio(3);
        // pop until an entry survives the distance cull (Core.h:63-68)
        uint32_t link;   // read only where `go` holds
        asm volatile("" : "=v"(link));   // no instruction: spares the v_mov of a 0 nobody reads
        bool go = false;
        if (MODE == 0) {
            // the top TWO entries in one LDS round trip (Core.h:63-68 pops until an entry survives the cull: a loop of
            // dependent LDS reads here, 1.4 of them per trip on average); an octet whose top two entries are both culled
            // drops them and sits this trip out. Same entries, same order, same results. Measured on the bench workload,
            // kernel alone: 24.0 ms per pass against 24.55 with the loop, and 25.0 with the loop kept for the third entry on.
            if (sp >= 0) {
                const unsigned long long e1 = s_stack[sp * SRX_OCTETS_PER_BLOCK + oct];
                const unsigned long long e2 = s_stack[(sp - 1) * SRX_OCTETS_PER_BLOCK + oct];
                const bool ok1 = !(u2f((uint32_t)(e1 >> 32)) >= best_t);
                const bool ok2 = sp >= 1 && !(u2f((uint32_t)(e2 >> 32)) >= best_t);
                link = ok1 ? (uint32_t)e1 : (uint32_t)e2;
                go = ok1 || ok2;
                sp -= ok1 ? 1 : 2;
                if (sp < -1) sp = -1;
            }
        } else if (sp >= 0) {
            // shadow rays: one pop per trip, no cull (above)
            link = (uint32_t)s_stack[sp * SRX_OCTETS_PER_BLOCK + oct];
            sp--;
            go = true;
        }
        if (go) {
            const bool leaf = (link & SRX_LINK_LEAF) != 0u;
            // node (2 lines) and packet (3 lines) fetches are issued together, before the two kinds part ways;
            // the shift drops the leaf bit: byte offset = line * 128 + lane * 16, off one SGPR base
            const uint32_t off = (link << 7) | sub16;
            // made before the loads go out (the empty asm pins it here): left to the compiler it is made in the leaf branch, in a
            // register the loads above write, and the third load then waits for the first two
            uint32_t off3 = (link << 7) + sub8;
            asm volatile("" : "+v"(off3));
            const char* const rec = sc.geom + (size_t)off;
            const float4 A = *(const float4*)rec;
#ifdef SRX_EXP_B3
            const float3 B3 = *(const float3*)(rec + 128);
            const float4 B = make_float4(B3.x, B3.y, B3.z, 0.0f);
#else
            const float4 B = *(const float4*)(rec + 128);   // {max.y, max.z, link, cone} / {e1.y, e1.z, e2.x, 0}
#endif
            const float4 r0 = ray_setup[0];                 // the ray's direction (and o.x): both kinds of trip need it
            const f3 d = UNI ? u_d : f3(r0.x, r0.y, r0.z);
            __builtin_amdgcn_s_setprio(0);
            if (!leaf) {
                n_nodes++;
                bool valid;
                unsigned long long bal;
                const uint32_t rb = fast ? node_test<true, UNI>(A, B, inv, fused, d, c1, dq, floor_t, zero_key_limit, valid, bal)
                                         : node_test<false, UNI>(A, B, inv, fused, d, c1, dq, floor_t, zero_key_limit, valid, bal);
                const uint32_t vmask = (uint32_t)(bal >> oct_shift) & 0xFFu;
                const int nvalid = __popc(vmask);
                int rank;
                if (MODE == 0) {
                    // total order of floats as unsigned ints (no NaN keys exist); invalid lanes sort last
                    // kb = (rb & ~7) | lane id (the key, Core.h:74-78), ord = kb ^ ((kb >> 31, arithmetic) | 0x80000000): as three
                    // instructions (bit-field insert, shift, one three-input bit operation: a ^ (b | c) is table 0x1e) where the
                    // compiler makes four
                    uint32_t kb, ok;
                    const uint32_t sign = (uint32_t)((int32_t)rb >> 31);
                    asm("v_bfi_b32 %0, 7, %1, %2" : "=v"(kb) : "v"(sub), "v"(rb));
                    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x1e" : "=v"(ok) : "v"(kb), "v"(sign), "s"(0x80000000u));
                    const uint32_t ord = valid ? ok : 0xFFFFFFFFu;
                    rank = octet_rank_u32(ord);
                } else {
                    // Shadow casts only ask "is there a hit with 0 < t <= 9999": the answer does not depend on the
                    // visiting order (a box the reference culls is culled behind a hit it already has): no sort.
                    rank = __popc(vmask & ((1u << sub) - 1u));
                }
                // push far -> near so the nearest child is popped first (Core.h:98-106); stored distance is the
                // unmodified tmin
                if (valid) s_stack[(sp + nvalid - rank) * SRX_OCTETS_PER_BLOCK + oct] = ((unsigned long long)rb << 32) | f2u(B.z);
                sp += nvalid;
            } else {
                // ---- leaf: one packet of 8 triangles per trip, one triangle per lane (Primitives.h:159-202).
                // The rest of a multi-packet leaf goes back on top with distance -inf: it is never culled, exactly
                // like the reference's inner `for` over the leaf's packets (Core.h:179).
                const float2 C = *(const float2*)(sc.geom + (size_t)off3);   // third line: 8 x {e2.y, e2.z}, 8-B slots
                const float2 r1 = *(const float2*)(ray_setup + 1);
                const f3 o(r0.w, r1.x, r1.y);
                if ((link & (SRX_LINK_MORE_MAX << SRX_LINK_MORE_SHIFT)) != 0u) {
                    sp++;   // next packet of the leaf: line + 3, one packet fewer to follow
                    if (sub == 0) s_stack[sp * SRX_OCTETS_PER_BLOCK + oct] = (0xFF800000ull << 32) | (link + 3u - (1u << SRX_LINK_MORE_SHIFT));
                }
                n_packets++;
                const float e1x = A.w, e1y = B.x, e1z = B.y;
                const float e2x = B.z, e2y = C.x, e2z = C.y;
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
                const bool cand = pass && (tt > 0.0f) && (MODE == 0 ? tt < best_t : tt <= 9999.0f);
                const unsigned long long cb = __builtin_amdgcn_ballot_w64(cand);
                if (cb != 0ull) {
                    if (MODE == 0) {
                        const uint32_t tb = cand ? f2u(tt) : 0xFFFFFFFFu;
                        const uint32_t m = octet_min_u32(tb);
                        if (m != 0xFFFFFFFFu) {
                            best_t = u2f(m);
                            const unsigned long long wb = __builtin_amdgcn_ballot_w64(tb == m);
                            const int w = __ffs((int)((uint32_t)(wb >> oct_shift) & 0xFFu)) - 1;
                            // the winning lane records (u, v, where its material and tri_index are) for its octet
                            if (sub == w) s_best[oct] = make_uint4(f2u(uu), f2u(vv), off + 256u + 64u - 12u * (uint32_t)sub, 0u);
                        }
                    } else if (((uint32_t)(cb >> oct_shift) & 0xFFu) != 0u) {
                        sp = -2;   // any occluder within 9999 => shadowed (Primitives.h:44-50); the octet's eight lanes agree
                        if (SPILL && spilled_mask != 0ull) s_spilled[oct] = 0u;
                    }
                }
            }
        }
        if (SPILL) {
            // Room for the next trip's push (one pop, up to 8 children) is made here, where the whole wave is together:
            // the bottom half of the LDS part goes to memory, the rest moves down. An octet whose LDS part is empty
            // while entries are parked takes the last half back. Neither happens on the scenes measured so far (the
            // bound is a worst case); the cost on every trip is one compare and two uniform branches.
            const bool over = sp > D - 8;
            const unsigned long long ob = __builtin_amdgcn_ballot_w64(over);
            if ((ob | spilled_mask) != 0ull) {
                const int H = D >> 1;
                uint32_t parked = s_spilled[oct];
                const bool under = sp < 0 && parked > 0u;
                if (ob != 0ull && spill_slot < 0) {
                    int got = -1;
                    if (lane == 0) {
                        for (int tries = 0; tries < (1 << 16) && got < 0; tries++) {
                            const uint32_t i = atomicAdd(&sc.spill_ctl[0], 1u) & (SRX_SPILL_SLOTS - 1u);
                            if (atomicCAS(&sc.spill_ctl[16 + i], 0u, 1u) == 0u) got = (int)i;
                        }
                        if (got < 0) atomicExch(&sc.spill_ctl[1], 1u);   // the host reports it; the rays concerned end early
                        else atomicAdd(&sc.spill_ctl[2], 1u);
                    }
                    spill_slot = __builtin_amdgcn_readfirstlane(got);
                }
                unsigned long long* const area = sc.spill_area + ((size_t)(spill_slot < 0 ? 0 : spill_slot) * SRX_OCTETS_PER_BLOCK + oct) * (size_t)sc.stack_depth;
                if (over) {
                    if (spill_slot < 0) {
                        sp = -1; parked = 0u;      // no area (cannot happen with SRX_SPILL_SLOTS > resident waves): flagged above
                    } else {
                        for (int i = sub; i < H; i += 8)
                            __hip_atomic_store(&area[parked + i], s_stack[i * SRX_OCTETS_PER_BLOCK + oct], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int i = sub; i + H <= sp; i += 8) {
                            const unsigned long long e = s_stack[(i + H) * SRX_OCTETS_PER_BLOCK + oct];
                            s_stack[i * SRX_OCTETS_PER_BLOCK + oct] = e;
                        }
                        sp -= H; parked += (uint32_t)H;
                    }
                } else if (under) {
                    for (int i = sub; i < H; i += 8)
                        s_stack[i * SRX_OCTETS_PER_BLOCK + oct] = __hip_atomic_load(&area[parked - H + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sp = H - 1; parked -= (uint32_t)H;
                }
                if (sub == 0) s_spilled[oct] = parked;
                spilled_mask = __builtin_amdgcn_ballot_w64(parked > 0u);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (K > 1) {
            // An octet whose ray has ended writes its result and takes the next ray of the wave's block that nobody
            // has taken yet (the K x 8 rays were all set up into LDS at the start): the wave ends when the block's rays
            // are used up, not when the octet with the most work of a fixed four ends.
            const bool ended = sp < 0 && cur >= 0;
            const unsigned long long eb = __builtin_amdgcn_ballot_w64(ended && sub == 0);
            if (eb != 0ull) {
                if (ended) {
                    finish(ray);
                    const int j = next_j + __popcll(eb & ((1ull << oct_shift) - 1ull));
                    cur = -1;
                    if (j < block_rays) {
                        cur = j;
                        ray = block_start + (uint32_t)j;
                        load_ray_j(j);
                        best_t = SRX_MISS_DISTANCE;
                        sp = 0;
                        if (sub == 0) s_stack[oct] = 0ull;
                    }
                }
                next_j += __popcll(eb);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (K == 1 && live) finish(ray);
    if (SPILL && spill_slot >= 0 && lane == 0) __hip_atomic_store(&sc.spill_ctl[16 + spill_slot], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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
}

static size_t trace_lds_bytes(int lds_depth, int k, bool spill) {
    return (size_t)(lds_depth + 1) * SRX_OCTETS_PER_BLOCK * sizeof(unsigned long long) + SRX_OCTETS_PER_BLOCK * sizeof(uint4) +
           (size_t)SRX_OCTETS_PER_BLOCK * k * (3 * sizeof(float4) + sizeof(float2)) + (spill ? SRX_OCTETS_PER_BLOCK * sizeof(uint32_t) : 0);
}

template <int K, bool SPILL>
static void launch_k(int mode, uint32_t blocks, size_t lds, const GScene& sc, const TraceArgs& a, hipStream_t stream) {
    if (mode == 0) hipLaunchKernelGGL((k_trace<0, K, SPILL>), dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
    else if (mode == 1) hipLaunchKernelGGL((k_trace<1, K, SPILL>), dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
    else if (mode == 2) hipLaunchKernelGGL((k_trace<2, K, SPILL>), dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
    else hipLaunchKernelGGL((k_trace<3, K, SPILL>), dim3(blocks), dim3(SRX_TRACE_THREADS), lds, stream, sc, a);
}

extern "C" hipError_t srx_launch_trace(int mode, const GScene& sc, const TraceArgs& args, hipStream_t stream) {
    if (args.n == 0) return hipSuccess;
    TraceArgs a = args;
    // the renderer's shadow rays of a scene with one sun all share its direction: the form that keeps it in scalar registers
    // (SRX_SHADOW_UNIFORM=0: the general form, for A/B runs and tests)
    if (mode == 2 && sc.n_suns == 1) {
        const char* e = getenv("SRX_SHADOW_UNIFORM");      // read per launch: tests switch it between renderers of one process
        if (!e || atoi(e) != 0) mode = 3;
    }
    // Rays per octet for a form of the kernel: the most whose LDS footprint lets 32 waves share a CU, fewer for a launch
    // that cannot fill the chip's 8 192 wave slots (256 CUs x 32) with that many (more waves, each a shorter serial
    // chain: the last generations of a pass are launches of a few thousand rays).
    static const size_t lds_budget = [] { const char* e = getenv("SRX_LDS_BUDGET"); return (size_t)(e ? atoi(e) : (int)SRX_LDS_PER_WAVE_FOR_FULL_OCCUPANCY); }();   // experiments
    static const int spill_depth = [] { const char* e = getenv("SRX_SPILL_DEPTH"); return e ? (atoi(e) & ~1) : SRX_SPILL_LDS_DEPTH; }();   // experiments
    auto rays_per_octet = [&](int lds_depth, bool spill_form) {
        int k = SRX_RAYS_PER_OCTET_MAX;
        while (k > 1 && trace_lds_bytes(lds_depth, k, spill_form) > lds_budget) k >>= 1;
        static const uint64_t rounds8 = [] { const char* e = getenv("SRX_K8_ROUNDS"); return (uint64_t)(e ? atoi(e) : 4); }();
        const uint64_t expected = (a.n_ptr && a.n_hint) ? std::min<uint64_t>(a.n_hint, a.n) : a.n;
        while (k > 1 && expected < (k == 8 ? rounds8 : 1ull) * 8192ull * SRX_OCTETS_PER_BLOCK * (uint64_t)k) k >>= 1;
        return k;
    };
    // Plain form (the tree's whole worst-case stack in LDS) or SPILL form (SRX_SPILL_LDS_DEPTH entries in LDS, the
    // overflow area behind them): for trees with a bound of at least SRX_SPILL_MIN_BOUND whichever carries more rays per
    // octet, and the SPILL form when the plain one cannot keep 32 waves per CU at all. SRX_LDS_STACK=<n> forces n entries (tests: the spill path on ordinary scenes).
    int depth = sc.stack_depth;
    bool spill = false;
    int k = rays_per_octet(depth, false);
    if (sc.spill_area && sc.stack_depth >= SRX_SPILL_MIN_BOUND) {
        const int ks = rays_per_octet(spill_depth, true);
        if (ks > k || trace_lds_bytes(depth, k, false) > lds_budget) { depth = spill_depth; spill = true; k = ks; }
    }
    if (const char* e = getenv("SRX_LDS_STACK")) {
        const int forced_depth = atoi(e) & ~1;
        if (sc.spill_area && forced_depth >= SRX_SPILL_MIN_LDS_DEPTH && forced_depth < sc.stack_depth) { depth = forced_depth; spill = true; }
        else if (forced_depth >= sc.stack_depth) { depth = sc.stack_depth; spill = false; }
        k = rays_per_octet(depth, spill);
    }
    a.lds_depth = depth;
    // blocks per chunk of the XCD map (swept 8-512: flat, +3.8 % against one contiguous eighth per XCD; SRX_XCD_CHUNK=0 = that form)
    static const uint32_t xcd_chunk = [] { const char* e = getenv("SRX_XCD_CHUNK"); const int v = e ? atoi(e) : 32; return (uint32_t)(v > 0 ? v : 0); }();
    a.xcd_chunk = xcd_chunk ? xcd_chunk : 0x7fffffffu;
    static const int forced = [] { const char* e = getenv("SRX_RAYS_PER_OCTET"); return e ? atoi(e) : 0; }();   // experiments
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) k = forced;
    // blocks per tier of the shrinking tail (swept 0-32 768, SRX_TAIL_BLOCKS: the closed diffuse box, whose passes are a handful of
    // launches with nothing beside them, gains 3 / 6 / 7 % at 2 048 / 8 192 / 16 384; the atrium, whose tails the shadow kernels on the
    // second stream fill anyway, loses 0.2 / 0.6 / 0.7 %)
    static const uint32_t tail_blocks = [] { const char* e = getenv("SRX_TAIL_BLOCKS"); return (uint32_t)(e ? atoi(e) : 4096); }();
    a.tail_blocks = tail_blocks;
    uint32_t blocks = k == 8 ? srx_trace_blocks_bound<8>(a.n, tail_blocks) : k == 4 ? srx_trace_blocks_bound<4>(a.n, tail_blocks)
                    : k == 2 ? srx_trace_blocks_bound<2>(a.n, tail_blocks) : srx_trace_blocks_bound<1>(a.n, tail_blocks);
    blocks = (blocks + 7u) & ~7u;   // multiple of 8 so the XCD map covers every logical block exactly once
    const size_t lds = trace_lds_bytes(depth, k, spill);
    if (spill) {
        if (k == 8) launch_k<8, true>(mode, blocks, lds, sc, a, stream);
        else if (k == 4) launch_k<4, true>(mode, blocks, lds, sc, a, stream);
        else if (k == 2) launch_k<2, true>(mode, blocks, lds, sc, a, stream);
        else launch_k<1, true>(mode, blocks, lds, sc, a, stream);
    } else {
        if (k == 8) launch_k<8, false>(mode, blocks, lds, sc, a, stream);
        else if (k == 4) launch_k<4, false>(mode, blocks, lds, sc, a, stream);
        else if (k == 2) launch_k<2, false>(mode, blocks, lds, sc, a, stream);
        else launch_k<1, false>(mode, blocks, lds, sc, a, stream);
    }
    return hipGetLastError();
}

// ---- self-test of the hand-written cross-lane code (srx_init, once per device)
// octet_rank_u32 is inline asm with the gfx950 wait states between a VALU write of VCC / of a DPP source and its read put
// in by hand (the assembler is not told about them inside an asm block), and octet_min_u32 leans on DPP controls: a
// compiler that schedules around the block differently, or a different chip, must not silently give wrong child orders.
// 256 octets of pseudo-random distinct keys (and the corner cases: equal high bits, 0xFFFFFFFF fillers) through both,
// compared on the host with the definition: rank = number of smaller keys in the octet, min = the smallest.
__global__ __launch_bounds__(64) void k_octet_selftest(const uint32_t* keys, uint32_t* rank_out, uint32_t* min_out) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const uint32_t k = keys[i];
    rank_out[i] = (uint32_t)octet_rank_u32(k);
    min_out[i] = octet_min_u32(k);
}
void srx_set_error(const std::string& s);   // srx_scene.hip
extern "C" int srx_trace_selftest() {
    const uint32_t n = 256 * 8;
    std::vector<uint32_t> keys(n), want_rank(n), want_min(n), got_rank(n), got_min(n);
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto next = [&] { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (uint32_t)(state >> 24); };
    for (uint32_t o = 0; o < n / 8; o++) {
        const int fillers = (int)(o % 9);                          // 0..8 lanes hold the "invalid" key
        for (uint32_t l = 0; l < 8; l++) {
            uint32_t k = next();
            if (o % 5 == 1) k = (k & 0xFFu) << 3;                  // small keys
            if (o % 5 == 2) k |= 0xFFFFFF00u;                      // keys that differ in the low byte only
            k = (k & ~7u) | l;                                     // distinct, as the traversal's keys are (lane id in the low bits)
            if ((int)l < fillers && o % 2) k = 0xFFFFFFFFu;        // invalid lanes sort last (equal keys: rank counts strictly smaller ones)
            keys[o * 8 + l] = k;
        }
        for (uint32_t l = 0; l < 8; l++) {
            uint32_t r = 0, m = 0xFFFFFFFFu;
            for (uint32_t j = 0; j < 8; j++) { r += keys[o * 8 + j] < keys[o * 8 + l]; m = std::min(m, keys[o * 8 + j]); }
            want_rank[o * 8 + l] = r; want_min[o * 8 + l] = m;
        }
    }
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, 3 * n * sizeof(uint32_t)) != hipSuccess) { srx_set_error("srx_init: self-test allocation failed"); return 1; }
    hipError_t e = hipMemcpy(d, keys.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_octet_selftest, dim3(n / 64), dim3(64), 0, 0, d, d + n, d + 2 * n); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(got_rank.data(), d + n, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(got_min.data(), d + 2 * n, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { srx_set_error(std::string("srx_init: self-test: ") + hipGetErrorString(e)); return 1; }
    if (got_rank != want_rank || got_min != want_min) {
        srx_set_error("srx_init: the cross-lane rank / min code of the traversal kernel gives wrong results on this device with this build "
                      "(hand-placed wait states in csrc/srx_trace.hip octet_rank_u32): refusing to run");
        return 1;
    }
    return 0;
}

// Diagnostics (host only, tests/test_host_cpu.py): the dealing of n rays to the blocks of the kernel form with k rays per octet,
// exactly as the kernel computes it. start / count take up to max_blocks entries; *n_blocks = blocks used, *bound = the
// grid size the launcher takes for a launch of up to n rays.
extern "C" int srx_trace_xcd_map(uint32_t nblk, uint32_t chunk, uint32_t* logical) {
    if ((nblk & 7u) || chunk == 0 || !logical) return 1;
    for (uint32_t b = 0; b < nblk; b++) logical[b] = srx_xcd_logical_block(b, nblk, chunk);
    return 0;
}
extern "C" int srx_trace_dealing(uint32_t n, int32_t k, uint32_t tail_blocks, uint32_t max_blocks, uint32_t* start, uint32_t* count,
                                 uint32_t* n_blocks, uint32_t* bound) {
    auto run = [&](auto tag) {
        constexpr int K = decltype(tag)::value;
        const TraceLayout L = srx_trace_layout<K>(n, tail_blocks);
        if (n_blocks) *n_blocks = L.blocks;
        if (bound) *bound = srx_trace_blocks_bound<K>(n, tail_blocks);
        for (uint32_t b = 0; b < L.blocks && b < max_blocks && start && count; b++) srx_trace_block_range<K>(L, n, b, start[b], count[b]);
    };
    if (k == 8) run(std::integral_constant<int, 8>());
    else if (k == 4) run(std::integral_constant<int, 4>());
    else if (k == 2) run(std::integral_constant<int, 2>());
    else if (k == 1) run(std::integral_constant<int, 1>());
    else return 1;
    return 0;
}
