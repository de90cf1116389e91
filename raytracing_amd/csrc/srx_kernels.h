// raytracing_amd/csrc/srx_kernels.h — kernel argument blocks and launchers shared by the .hip units.
#pragma once
#include "srx_device.h"

struct TraceArgs {
    const float* org;  int org_stride;     // floats between consecutive ray origins
    const float* dir;  int dir_stride;     // unused by the shadow-queue form
    uint32_t n;
    GHit* hits;                            // closest form
    float* advanced_org3;                  // optional: o + t*d*0.999f (Core.h:327)
    uint8_t* lit;                          // shadow form, API
    const GShadow* shadow_q;               // shadow form, renderer queue
    float* sample3;                        // shadow form, renderer: per-pixel sample sums (atomic)
    unsigned long long* counters;          // [0] node tests, [1] packet tests
};

// mode 0: closest hit; 1: shadow (API form, writes lit[]); 2: shadow (renderer queue form, adds lit contributions)
extern "C" hipError_t srx_launch_trace(int mode, const GScene& sc, const TraceArgs& a, hipStream_t stream);
