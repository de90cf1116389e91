// raytracing_amd/csrc/srx_host.h — host-side internals shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/srx.h"
#include "srx_kernels.h"

void srx_set_error(const std::string& s);

#define SRX_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            srx_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                           \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

#define SRX_FAIL(msg)                 \
    do {                              \
        srx_set_error(msg);           \
        return 1;                     \
    } while (0)

// reference record layouts (byte offsets from SURVEY §8a: a2, a5, a11)
enum { SRX_REF_NODE_BYTES = 288, SRX_REF_PACKET_BYTES = 448, SRX_REF_TRI_BYTES = 196 };

int srx_scene_check_spill(const struct srx_scene* scene);   // srx_scene.hip: 1 (+ error text) if a traversal wave found no overflow area

struct srx_scene {
    GScene g;                         // device pointers + parameters, passed by value to kernels
    int device = 0;
    std::vector<void*> allocations;   // every hipMalloc owned by this scene
    std::vector<uint8_t> tri_material_host;
};

