// raytracing_amd/csrc/srx_deal.hip — C-ABI: noise-weighted dealing of the frame's row groups to the ranks of a job.
//
// The reference re-deals its 4x4 blocks to its render threads after every noise analysis so that the threads' burdens
// (the sums of their blocks' priorities) come within 5 % of each other: SceneManager::balance_threads, second half
// (SceneManager.h:693-773: blocks move from the FRONT of the most burdened thread's list to the BACK of the least
// burdened one's until the two are within 5 %, at most thread_count rounds). Across GPUs the unit that is dealt is a group
// of pixel rows (srx_trace_pass_interleaved's group_rows, a multiple of the block size) and a part is a rank; the
// algorithm is the reference's, restated on those. Everything here is a function of the block priorities alone, which
// every rank computes identically (srx_noise_priorities on the complete error map), so every rank arrives at the same
// deal without a collective. A deal starts as the round-robin of srx_trace_pass_interleaved
// (SceneManager::assign_iterative_groups, :441-459).
// Host code except for the device mirror of the tables and the pack / unpack kernels of the dealt gathers.
#include "srx_host.h"
#include <algorithm>
#include <cmath>

namespace {
void renumber(srx_deal* d) {
    // a short last group of the frame goes last in its owner's list (local row j -> group j / group_rows needs every
    // earlier group of the list to be whole)
    if (d->res_y % d->group) {
        const int last = d->n_groups - 1;
        std::vector<int>& L = d->lists[d->owner[last]];
        auto it = std::find(L.begin(), L.end(), last);
        if (it != L.end()) { L.erase(it); L.push_back(last); }
    }
    d->list_off.assign(d->nparts + 1, 0);
    for (int p = 0; p < d->nparts; p++) {
        for (size_t k = 0; k < d->lists[p].size(); k++) { d->owner[d->lists[p][k]] = p; d->slot[d->lists[p][k]] = (int)k; }
        d->list_off[p + 1] = d->list_off[p] + (int)d->lists[p].size();
    }
    d->version++;
}
}  // namespace

int srx_deal_rows_of_part(const srx_deal* d, int part) {
    int n = 0;
    for (int g : d->lists[part]) n += std::min(d->group, d->res_y - g * d->group);
    return n;
}
int srx_deal_most_rows(const srx_deal* d) {
    int m = 0;
    for (int p = 0; p < d->nparts; p++) m = std::max(m, srx_deal_rows_of_part(d, p));
    return m;
}

extern "C" int srx_deal_create(int32_t res_x, int32_t res_y, int32_t group_rows, int32_t nparts, srx_deal** out) {
    if (!out) SRX_FAIL("srx_deal_create: null argument");
    if (res_x <= 0 || res_y <= 0 || group_rows <= 0 || nparts <= 0) SRX_FAIL("srx_deal_create: bad frame or row interleave (res_x, res_y, group_rows, nparts)");
    srx_deal* d = new srx_deal();
    d->res_x = res_x; d->res_y = res_y; d->group = group_rows; d->nparts = nparts;
    d->n_groups = (res_y + group_rows - 1) / group_rows;
    d->owner.assign(d->n_groups, 0); d->slot.assign(d->n_groups, 0); d->cost.assign(d->n_groups, 0.0f);
    d->lists.assign(nparts, std::vector<int>());
    for (int g = 0; g < d->n_groups; g++) d->lists[g % nparts].push_back(g);     // SceneManager.h:441-459
    renumber(d);
    *out = d;
    return 0;
}

extern "C" void srx_deal_destroy(srx_deal* d) {
    if (!d) return;
    if (d->d_tables) (void)hipFree(d->d_tables);
    delete d;
}

// SceneManager.h:693-773 on row groups. block_priority: HOST array, one float per 4x4 block, row-major over
// ceil(res_x / 4) x ceil(res_y / 4) (what srx_noise_priorities wrote, copied back by the caller).
extern "C" int srx_deal_rebalance(srx_deal* d, const float* block_priority, uint32_t n_blocks, int32_t* groups_moved) {
    if (!d || !block_priority) SRX_FAIL("srx_deal_rebalance: null argument");
    if (d->group % 4) SRX_FAIL("srx_deal_rebalance: group_rows must be a multiple of the 4-pixel block size (SceneManager.h:299)");
    const int bx = (d->res_x + 3) / 4, by = (d->res_y + 3) / 4;
    if (n_blocks != (uint32_t)(bx * by)) SRX_FAIL("srx_deal_rebalance: n_blocks does not match the frame");
    // a group's burden: the priorities of its blocks, summed in block order (:699-706 sums a thread's blocks the same way)
    for (int g = 0; g < d->n_groups; g++) {
        float c = 0.0f;
        for (int r = g * d->group / 4; r < std::min(by, (g + 1) * d->group / 4); r++)
            for (int x = 0; x < bx; x++) c += block_priority[(size_t)r * bx + x];
        d->cost[g] = c;
    }
    const int T = d->nparts;
    std::vector<float> burden(T, 0.0f);
    for (int p = 0; p < T; p++) for (int g : d->lists[p]) burden[p] += d->cost[g];
    int moved = 0;
    float similarity = 0.0f;
    int iterations = 0;
    int moved_before;
    do {                                                                          // :708-762
        iterations++;
        moved_before = moved;
        float max_burden = burden[0], min_burden = burden[0];
        int max_id = 0, min_id = 0;
        for (int t = 1; t < T; t++) {
            if (burden[t] > max_burden) { max_burden = burden[t]; max_id = t; }
            if (burden[t] < min_burden) { min_burden = burden[t]; min_id = t; }
        }
        similarity = (max_burden - min_burden) / max_burden;
        std::vector<int>& from = d->lists[max_id];
        std::vector<int>& to = d->lists[min_id];
        // :730-741 moves the block at the FRONT of the list whatever its burden. A row group carries the burden of a whole
        // row of blocks, so one group can outweigh the gap and the pair would trade it back and forth: here the first group
        // from the front whose burden is smaller than the gap moves (the gap then shrinks strictly), and the round ends when
        // there is none; a group without burden stays where it is (moving it changes nothing but the size of the shares the
        // gathers pad to). (The reference has no guard on an empty list either; with NaN or zero burdens the comparison is
        // false and nothing moves, there as here.)
        while (similarity > 0.05f && max_id != min_id && from.size() > 1) {
            size_t k = 0;
            while (k < from.size() && !(d->cost[from[k]] > 0.0f && d->cost[from[k]] < max_burden - min_burden)) k++;
            if (k == from.size()) break;
            const int g = from[k];
            const float b = d->cost[g];
            max_burden -= b; min_burden += b;
            from.erase(from.begin() + (long)k); to.push_back(g);
            moved++;
            similarity = (max_burden - min_burden) / max_burden;
        }
        burden[min_id] = min_burden; burden[max_id] = max_burden;
        // :749-758 carries the pair's burdens into the scan for the next extremes (after an overshoot the "max" it carries is
        // the smaller of the two); the extremes are taken afresh here
        max_burden = burden[0]; min_burden = burden[0];
        for (int t = 1; t < T; t++) {
            if (burden[t] > max_burden) max_burden = burden[t];
            if (burden[t] < min_burden) min_burden = burden[t];
        }
        similarity = (max_burden - min_burden) / max_burden;
    } while (similarity > 0.05f && moved > moved_before && iterations < 64 * T);   // the reference: thread_count rounds; here until a round moves nothing
    renumber(d);
    if (groups_moved) *groups_moved = moved;
    return 0;
}

extern "C" int srx_deal_get(const srx_deal* d, int32_t* owner, int32_t* slot, uint32_t capacity, uint32_t* n_groups, float* burdens) {
    if (!d) SRX_FAIL("srx_deal_get: null deal");
    if (n_groups) *n_groups = (uint32_t)d->n_groups;
    for (uint32_t g = 0; g < capacity && g < (uint32_t)d->n_groups; g++) {
        if (owner) owner[g] = d->owner[g];
        if (slot) slot[g] = d->slot[g];
    }
    if (burdens)
        for (int p = 0; p < d->nparts; p++) { float b = 0.0f; for (int g : d->lists[p]) b += d->cost[g]; burdens[p] = b; }
    return 0;
}

int srx_deal_device(srx_deal* d, int part, const int** list, const int** owner, const int** slot) {
    if (part < 0 || part >= d->nparts) SRX_FAIL("srx_deal: part out of range");
    int dev = -1;
    SRX_HIP(hipGetDevice(&dev));
    if (d->d_tables && d->d_device != dev) { (void)hipFree(d->d_tables); d->d_tables = nullptr; }
    if (!d->d_tables) { SRX_HIP(hipMalloc((void**)&d->d_tables, (size_t)3 * d->n_groups * sizeof(int))); d->d_device = dev; d->d_version = 0; }
    if (d->d_version != d->version) {
        std::vector<int> h((size_t)3 * d->n_groups);
        std::copy(d->owner.begin(), d->owner.end(), h.begin());
        std::copy(d->slot.begin(), d->slot.end(), h.begin() + d->n_groups);
        for (int p = 0; p < d->nparts; p++) std::copy(d->lists[p].begin(), d->lists[p].end(), h.begin() + 2 * d->n_groups + d->list_off[p]);
        SRX_HIP(hipDeviceSynchronize());                      // no kernel of an earlier call may still read the old tables
        SRX_HIP(hipMemcpy(d->d_tables, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
        d->d_version = d->version;
    }
    if (owner) *owner = d->d_tables;
    if (slot) *slot = d->d_tables + d->n_groups;
    if (list) *list = d->d_tables + 2 * d->n_groups + d->list_off[part];
    return 0;
}

// ---- the row packing of the gathers under a deal (csrc/srx_group.hip uses these; exported for hosts with their own transport)
namespace {
__global__ __launch_bounds__(256) void k_rows_pack_dealt(const float* src, int fpp, int res_x, int group, const int* list, int own_rows, float* slot) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_floats = (size_t)res_x * fpp;
    if (i >= (size_t)own_rows * row_floats) return;
    const int j = (int)(i / row_floats);
    const int y = list[j / group] * group + (j % group);
    slot[i] = src[(size_t)y * row_floats + (i - (size_t)j * row_floats)];
}
__global__ __launch_bounds__(256) void k_rows_unpack_dealt(const float* slots, size_t slot_stride, size_t buffer_offset, int fpp, int res_x, int res_y, int group,
                                                          const int* owner, const int* slot_of, int skip_part, float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_floats = (size_t)res_x * fpp;
    if (i >= (size_t)res_y * row_floats) return;
    const int y = (int)(i / row_floats);
    const int g = y / group, part = owner[g];
    if (part == skip_part) return;
    const int j = slot_of[g] * group + (y % group);
    dst[i] = slots[(size_t)part * slot_stride + buffer_offset + (size_t)j * row_floats + (i - (size_t)y * row_floats)];
}
}  // namespace

extern "C" uint64_t srx_deal_slot_floats(const srx_deal* d, int32_t floats_per_pixel) {
    if (!d || floats_per_pixel <= 0) return 0;
    return (uint64_t)srx_deal_most_rows(d) * (uint64_t)d->res_x * (uint64_t)floats_per_pixel;
}

extern "C" int srx_rows_pack_dealt(srx_deal* d, const float* d_frame, int32_t floats_per_pixel, int32_t part, float* d_slot, void* stream) {
    if (!d || !d_frame || !d_slot) SRX_FAIL("srx_rows_pack_dealt: null argument");
    if (floats_per_pixel <= 0 || floats_per_pixel > 16) SRX_FAIL("srx_rows_pack_dealt: bad floats per pixel");
    const int* list = nullptr;
    if (srx_deal_device(d, part, &list, nullptr, nullptr)) return 1;
    const int own = srx_deal_rows_of_part(d, part);
    const size_t n = (size_t)own * d->res_x * floats_per_pixel;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_rows_pack_dealt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_frame, floats_per_pixel, d->res_x, d->group, list,
                       own, d_slot);
    SRX_HIP(hipGetLastError());
    return 0;
}

extern "C" int srx_rows_unpack_dealt(srx_deal* d, const float* d_slots, uint64_t slot_stride_floats, uint64_t buffer_offset_floats, int32_t floats_per_pixel,
                                     int32_t skip_part, float* d_frame, void* stream) {
    if (!d || !d_slots || !d_frame) SRX_FAIL("srx_rows_unpack_dealt: null argument");
    if (floats_per_pixel <= 0 || floats_per_pixel > 16) SRX_FAIL("srx_rows_unpack_dealt: bad floats per pixel");
    const int *owner = nullptr, *slot = nullptr;
    if (srx_deal_device(d, 0, nullptr, &owner, &slot)) return 1;
    const size_t n = (size_t)d->res_y * d->res_x * floats_per_pixel;
    hipLaunchKernelGGL(k_rows_unpack_dealt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_slots, (size_t)slot_stride_floats,
                       (size_t)buffer_offset_floats, floats_per_pixel, d->res_x, d->res_y, d->group, owner, slot, skip_part, d_frame);
    SRX_HIP(hipGetLastError());
    return 0;
}
