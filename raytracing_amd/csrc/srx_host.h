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


// Noise-weighted dealing of the frame's row groups to the parts (ranks) of a job: csrc/srx_deal.hip.
struct srx_deal {
    int res_x = 0, res_y = 0, group = 0, nparts = 0, n_groups = 0;
    std::vector<int> owner;                 // [n_groups]
    std::vector<int> slot;                  // [n_groups]: index of the group in its owner's list
    std::vector<std::vector<int>> lists;    // [nparts]: a part's groups, in its own order
    std::vector<float> cost;                // [n_groups]: the burden of each group at the last rebalance (sum of its blocks' priorities)
    uint32_t version = 1;
    // device mirror, made on first use by a GPU call and refreshed after a rebalance: owner | slot | lists (concatenated)
    int* d_tables = nullptr;
    int d_device = -1;
    uint32_t d_version = 0;
    std::vector<int> list_off;              // [nparts + 1]: offsets of the lists in the concatenation
};
int srx_deal_rows_of_part(const srx_deal* d, int part);     // pixel rows a part owns
int srx_deal_most_rows(const srx_deal* d);                  // the largest of those
// device tables for `part` on the current device (uploads if stale): GRows fields list / owner, and the slot table
int srx_deal_device(srx_deal* d, int part, const int** list, const int** owner, const int** slot);
