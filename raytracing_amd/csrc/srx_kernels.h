// raytracing_amd/csrc/srx_kernels.h — kernel argument blocks and launchers shared by the .hip units.
#pragma once
#include "srx_device.h"

struct TraceArgs {
    const float* org;  int org_stride;     // floats between consecutive ray origins
    const float* dir;  int dir_stride;     // unused by the shadow-queue form
    uint32_t n;
    const uint32_t* n_ptr;                 // optional (device): the launch processes min(*n_ptr, n) rays; n sizes the grid (launches
                                           // enqueued before the host knows how many rays the previous kernel produced)
    uint32_t n_hint;                       // with n_ptr: how many rays the host expects (0 = n); only picks the rays per octet
    GHit* hits;                            // closest form
    float* advanced_org3;                  // optional: o + t*d*0.999f (Core.h:327)
    uint8_t* lit;                          // shadow form, API
    const GShadow* shadow_q;               // shadow form, renderer queue
    float* sample3;                        // shadow form, renderer: per-pixel sample sums (atomic)
    unsigned long long* counters;          // [0] node tests, [1] packet tests (of slot 0)
    uint32_t counter_slot_mask;            // counters are replicated in (mask+1) slots to spread the atomics ...
    uint32_t counter_slot_stride;          // ... slot s starts at counters + s*stride (u64 units); host sums the slots
    int lds_depth;                         // set by srx_launch_trace: stack entries per octet kept in LDS (<= GScene::stack_depth)
    uint32_t xcd_chunk;                    // set by srx_launch_trace: blocks per chunk of the XCD-aware block -> ray-range map
    uint32_t tail_blocks;                  // set by srx_launch_trace: blocks per tier of the shrinking tail (srx_trace_tiers), 0 = none
};

// How the rays of a launch are dealt to its blocks (one wave each). A block of the kernel form with K rays per octet can
// hold 8 K rays; dealing every block that many makes the launch end with its slowest wave still working through its pool
// while the chip drains. So the END of the ray range is dealt in smaller portions: tier t = 1, 2, ... log2 K holds up to
// `tail_blocks` blocks of 8 K >> t rays, the last one 8 rays (one per octet); tier 0 takes everything before them in full
// blocks. Blocks are dispatched in index order, so the waves that start last have the least to do (guided
// self-scheduling). Tier boundaries are multiples of a full block counted back from the end of the range rounded up to
// one, so every tier but the last holds whole blocks. Results do not depend on the dealing.
// All scalars (no arrays: the kernel keeps them in SGPRs): first ray fr[1..3] and first block fb[1..3] of tiers 1..3
// (tier 0 starts at ray 0, block 0); returns the number of blocks.
struct TraceLayout { uint32_t fr1, fr2, fr3, fb1, fb2, fb3, blocks; };
template <int K>
__host__ __device__ inline TraceLayout srx_trace_layout(uint32_t n, uint32_t tail_blocks) {
    constexpr int LEVELS = K == 8 ? 3 : (K == 4 ? 2 : (K == 2 ? 1 : 0));
    constexpr uint32_t FULL = 8u * K;
    const uint32_t tb = tail_blocks - tail_blocks % (uint32_t)K;
    TraceLayout L = {0, 0, 0, 0, 0, 0, 0};
    const uint64_t end = ((uint64_t)n + FULL - 1) / FULL * FULL;
    uint64_t at = end;
    if (LEVELS >= 3) { const uint64_t cap = (uint64_t)tb * (FULL >> 3); at = at > cap ? at - cap : 0; L.fr3 = (uint32_t)at; }
    if (LEVELS >= 2) { const uint64_t cap = (uint64_t)tb * (FULL >> 2); at = at > cap ? at - cap : 0; L.fr2 = (uint32_t)at; }
    if (LEVELS >= 1) { const uint64_t cap = (uint64_t)tb * (FULL >> 1); at = at > cap ? at - cap : 0; L.fr1 = (uint32_t)at; }
    auto between = [n](uint32_t from, uint32_t stop, uint32_t size) {      // blocks of `size` rays over [from, stop), cut at n
        from = from < n ? from : n; stop = stop < n ? stop : n;
        return stop > from ? (stop - from + size - 1u) / size : 0u;
    };
    const uint32_t e0 = LEVELS >= 1 ? L.fr1 : n, e1 = LEVELS >= 2 ? L.fr2 : n, e2 = LEVELS >= 3 ? L.fr3 : n;
    L.fb1 = between(0u, e0, FULL);
    L.fb2 = L.fb1 + (LEVELS >= 1 ? between(L.fr1, e1, FULL >> 1) : 0u);
    L.fb3 = L.fb2 + (LEVELS >= 2 ? between(L.fr2, e2, FULL >> 2) : 0u);
    L.blocks = L.fb3 + (LEVELS >= 3 ? between(L.fr3, n, FULL >> 3) : 0u);
    return L;
}
// the rays of logical block b: [start, start + count)
template <int K>
__host__ __device__ inline void srx_trace_block_range(const TraceLayout& L, uint32_t n, uint32_t b, uint32_t& start, uint32_t& count) {
    constexpr int LEVELS = K == 8 ? 3 : (K == 4 ? 2 : (K == 2 ? 1 : 0));
    constexpr uint32_t FULL = 8u * K;
    uint32_t size = FULL, first_ray = 0, first_block = 0;
    if (LEVELS >= 1 && b >= L.fb1) { size = FULL >> 1; first_ray = L.fr1; first_block = L.fb1; }
    if (LEVELS >= 2 && b >= L.fb2) { size = FULL >> 2; first_ray = L.fr2; first_block = L.fb2; }
    if (LEVELS >= 3 && b >= L.fb3) { size = FULL >> 3; first_ray = L.fr3; first_block = L.fb3; }
    start = first_ray + (b - first_block) * size;
    count = (b < L.blocks && start < n) ? ((n - start) < size ? (n - start) : size) : 0u;
}
// an upper bound of the block count for every ray count <= n (the count itself dips by up to K - 1 each time the rounded
// end moves): what a grid must have when the launch reads its ray count on the device
template <int K>
__host__ __device__ inline uint32_t srx_trace_blocks_bound(uint32_t n, uint32_t tail_blocks) {
    const uint64_t end = ((uint64_t)n + 8u * K - 1) / (8u * K) * (8u * K);
    return srx_trace_layout<K>((uint32_t)(end > 0xFFFFFFFFull ? 0xFFFFFFFFull : end), tail_blocks).blocks;
}

// XCD-aware placement: hardware block b runs on XCD b % 8; XCD x works through chunks x, x + 8, x + 16, ... of `chunk`
// consecutive logical blocks (what is left over when the blocks per XCD are not a multiple of `chunk` is dealt as one
// shorter chunk per XCD at the end). nblk is a multiple of 8; a bijection of [0, nblk).
__host__ __device__ inline uint32_t srx_xcd_logical_block(uint32_t b, uint32_t nblk, uint32_t chunk) {
    const uint32_t per_xcd = nblk >> 3, x = b & 7u, j = b >> 3;
    const uint32_t full = per_xcd / chunk, rem = per_xcd - full * chunk;
    return j < full * chunk ? ((j / chunk) * 8u + x) * chunk + j % chunk : full * chunk * 8u + x * rem + (j - full * chunk);
}

// mode 0: closest hit; 1: shadow (API form, writes lit[]); 2: shadow (renderer queue form, adds lit contributions)
extern "C" hipError_t srx_launch_trace(int mode, const GScene& sc, const TraceArgs& a, hipStream_t stream);
extern "C" int srx_trace_selftest();   // 0 = the cross-lane rank / min code gives the defined results on the current device

// == srx_camera (include/srx.h), Camera.h:32-140
struct GCamera {
    float pos[3]; float rot[4]; float focal[3];
    float lens_width, lens_height;
    int res_x, res_y, subdiv;
};

// A parked bounce loop (ShadeArgs::cont_out): the ray's slot in the segment and the loop's state (Core.h:498-591: slots
// used, attempts made, the throughput carried into a retry, quirk Q8) with the ray's random stream.
struct GCont { uint32_t idx, i, ri; float tr, tg, tb; uint64_t s; };
struct ShadeArgs {
    const GRay* rays;              // queue segment being shaded
    const GHit* hits;              // hit record of each ray of the segment
    uint32_t n;
    const uint32_t* n_ptr;         // optional (device): min(*n_ptr, n) rays, as in TraceArgs
    int generation;                // PackagedRay::generation of every ray in the segment (Ray.h:10)
    GRay* out_rays;                // next generation's rays are appended here ...
    uint32_t* out_count;           // ... at *out_count (device counter, one atomic per wave)
    GShadow* shadow_q;             // shadow rays of this segment (one per hit and sun light)
    uint32_t* shadow_count;
    float* sample3;                // per-pixel sample sums (float atomics)
    unsigned long long* counters;  // [4] hits, [5] texels, [6] pdf lookups, [7] children (of slot 0)
    uint32_t counter_slot_mask, counter_slot_stride;   // as in TraceArgs
    // Rounds (optional): a wave whose bounce loop is down to `spill_below` busy lanes or fewer parks them in `cont_out`
    // (one GCont each) and leaves; a later launch with `cont_in` set takes min(*n_ptr, n) RECORDS, 64 to a wave again, and
    // goes on where they stopped (no emissive term, no shadow rays, no hit / texel counts: the first round made those).
    const GCont* cont_in;
    GCont* cont_out;
    uint32_t* cont_count;          // device counter of cont_out
    uint32_t cont_capacity;        // slots of cont_out (the launcher sizes it so that it cannot overflow)
    int spill_below;
    int spill_after;               // ... and not before the wave has made this many trips
    // Optional: the launch's units are the entries of this list of slots of `rays` / `hits` (srx_launch_hitlist: the rays of
    // a batch that hit something, so that an open scene's misses do not leave lanes idle); n / n_ptr then count entries.
    const uint32_t* slot_list;
};

// Writes the slots (0 .. min(*n_ptr, n) - 1) of the rays that hit something to `list` and their number to *count (one
// atomic per block of 1 024 rays; order: by block as the blocks get their turn, ascending within a block).
extern "C" hipError_t srx_launch_hitlist(const GHit* hits, uint32_t n, const uint32_t* n_ptr, uint32_t* list, uint32_t* count, hipStream_t stream);

extern "C" hipError_t srx_launch_shade(const GScene& sc, const ShadeArgs& a, hipStream_t stream);
// Rows of a pass: local row j of this caller is image row
//   y0 + ((j / group) * nparts + part) * group + j % group
// i.e. groups of `group` rows dealt round-robin over `nparts` callers (nparts == 1: the plain rectangle).
// With a deal (srx_deal, noise-weighted dealing of the row groups): `list` (device) = this part's groups in its own order
// (a short last group of the frame is always last in its owner's list), `owner` (device) = the part every group belongs to;
// local row j is then image row y0 + list[j / group] * group + j % group.
struct GRows { int y0, group, nparts, part; const int* list = nullptr; const int* owner = nullptr; };
__host__ __device__ inline int srx_row_of(const GRows& m, int j) {
    if (m.list) return m.y0 + m.list[j / m.group] * m.group + (j % m.group);
    return m.y0 + ((j / m.group) * m.nparts + m.part) * m.group + (j % m.group);
}
__host__ __device__ inline bool srx_row_owned(const GRows& m, int y) {
    const int rel = y - m.y0;
    if (rel < 0) return false;
    return m.owner ? m.owner[rel / m.group] == m.part : ((rel / m.group) % m.nparts) == m.part;
}
// rays for local rows [j0, j0 + rows) x columns [x0, x0 + w)
extern "C" hipError_t srx_launch_raygen(const GCamera& c, int x0, int w, GRows rows_map, int j0, int rows, uint32_t pass_index,
                                        uint64_t seed, uint32_t slot_base, GRay* rays, float* dir3, unsigned long long* keys,
                                        hipStream_t stream);
// Adaptive visit (SceneManager.h:193-237): rays only for the pixels of blocks that fire; appended compactly at rays[*count].
struct AdaptiveArgs {
    const float* block_priority;     // per 4x4 block, NULL = all 1
    const float4* accum4;            // the pixel's own sample count (4th float) is its sample index
    uint32_t visit_index;
    uint64_t seed;
    uint32_t blocks_x, n_blocks;
};
extern "C" hipError_t srx_launch_raygen_adaptive(const GCamera& c, int x0, int w, GRows rows_map, int j0, int rows, AdaptiveArgs a,
                                                 GRay* rays, uint32_t* count, hipStream_t stream);
extern "C" hipError_t srx_launch_accumulate_adaptive(int res_x, int x0, int w, GRows rows_map, int rows, AdaptiveArgs a, float* sample3,
                                                     float* accum4, float* aux4, hipStream_t stream);
extern "C" hipError_t srx_launch_pack_rays(const float* org3, const float* dir3, const unsigned long long* keys, uint32_t n,
                                           GRay* rays, hipStream_t stream);
extern "C" hipError_t srx_launch_accumulate(int res_x, int x0, int w, GRows rows_map, int rows, uint32_t pass_index, int n_passes,
                                            uint32_t slot_stride, float* sample3, float* accum4, float* aux4, hipStream_t stream);
// the reference's own generator on the device (XorRandom.cpp): primaries of a rectangle from ONE sequential XorGen stream, and raw streams
extern "C" hipError_t srx_launch_raygen_xorgen(const GCamera& c, int x0, int y0, int w, int h, uint32_t pass_index, const uint64_t* seed4, float* dir3,
                                               hipStream_t stream);
extern "C" hipError_t srx_launch_xorgen_stream(const uint64_t* seed4, uint32_t n_pools, uint32_t n_scalar, float* pools8, float* scalars, hipStream_t stream);
