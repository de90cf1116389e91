// raytracing_amd/csrc/srx_kernels.h — kernel argument blocks and launchers shared by the .hip units.
#pragma once
#include "srx_device.h"

struct TraceArgs {
    const float* org;  int org_stride;     // floats between consecutive ray origins
    const float* dir;  int dir_stride;     // unused by the shadow-queue form
    uint32_t n;
    const uint32_t* n_ptr;                 // optional (device): the launch processes min(*n_ptr, n) rays; n sizes the grid (launches
                                           // enqueued before the host knows how many rays the previous kernel produced)
    GHit* hits;                            // closest form
    float* advanced_org3;                  // optional: o + t*d*0.999f (Core.h:327)
    uint8_t* lit;                          // shadow form, API
    const GShadow* shadow_q;               // shadow form, renderer queue
    float* sample3;                        // shadow form, renderer: per-pixel sample sums (atomic)
    unsigned long long* counters;          // [0] node tests, [1] packet tests (of slot 0)
    uint32_t counter_slot_mask;            // counters are replicated in (mask+1) slots to spread the atomics ...
    uint32_t counter_slot_stride;          // ... slot s starts at counters + s*stride (u64 units); host sums the slots
    int lds_depth;                         // set by srx_launch_trace: stack entries per octet kept in LDS (<= GScene::stack_depth)
};

// mode 0: closest hit; 1: shadow (API form, writes lit[]); 2: shadow (renderer queue form, adds lit contributions)
extern "C" hipError_t srx_launch_trace(int mode, const GScene& sc, const TraceArgs& a, hipStream_t stream);

// == srx_camera (include/srx.h), Camera.h:32-140
struct GCamera {
    float pos[3]; float rot[4]; float focal[3];
    float lens_width, lens_height;
    int res_x, res_y, subdiv;
};

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
};

extern "C" hipError_t srx_launch_shade(const GScene& sc, const ShadeArgs& a, hipStream_t stream);
// Rows of a pass: local row j of this caller is image row
//   y0 + ((j / group) * nparts + part) * group + j % group
// i.e. groups of `group` rows dealt round-robin over `nparts` callers (nparts == 1: the plain rectangle).
struct GRows { int y0, group, nparts, part; };
__host__ __device__ inline int srx_row_of(const GRows& m, int j) {
    return m.y0 + ((j / m.group) * m.nparts + m.part) * m.group + (j % m.group);
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
