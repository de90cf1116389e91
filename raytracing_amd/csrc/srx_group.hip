// raytracing_amd/csrc/srx_group.hip — C-ABI: the multi-GPU exchange steps, one process per GPU.
//
// The reference is one process whose worker threads share memory; what they hand to each other there becomes a
// collective here (RCCL over xGMI):
//   SceneManager::get_screen_update   (SceneManager.h:801-815: every block's pixels copied into one output)
//        -> srx_group_gather_framebuffer: all-gather of each rank's tile rows, main and aux buffer in ONE collective
//   SceneManager::gather_thread_stats (:488-502: minimum iterations_done over the blocks) and the per-pixel error
//   terms analyze_noise sums over regions (:559-579)
//        -> srx_group_allgather_rows on the error map (each rank's rows travel once; the analysis then runs on the
//           complete map exactly as on one GPU) and srx_group_allreduce for the minimum and for counters.
// The row packing is exported on its own (srx_rows_pack / srx_rows_unpack) for hosts that bring their own transport
// (MPI, torch.distributed): raytracing_amd/tiles.py uses it that way.
// RCCL is loaded at run time (dlopen "librccl.so.1"): a process that already holds an RCCL (PyTorch's) shares it, and
// single-GPU users of the library do not need it at all.
#include "srx_host.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>

namespace {

// ---- the part of the RCCL API this file uses (rccl.h: ncclResult_t 0 = success, ncclUniqueId = 128 bytes)
typedef struct { char internal[128]; } rccl_unique_id;
typedef void* rccl_comm;
enum { RCCL_INT32 = 2, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8 };      // ncclInt32, ncclFloat32, ncclFloat64
enum { RCCL_SUM = 0, RCCL_MIN = 3 };                              // ncclSum, ncclMin
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) SRX_FAIL(std::string("srx_group: cannot load RCCL (librccl.so.1): ") + dlerror());
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(rccl_comm*, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (int (*)(rccl_comm))dlsym(h, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, rccl_comm, hipStream_t))dlsym(h, "ncclAllGather");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString)
        SRX_FAIL("srx_group: the RCCL library lacks a symbol this library needs");
    g_rccl = r;
    return 0;
}

#define SRX_RCCL(call)                                                                                              \
    do {                                                                                                            \
        int rc_ = (call);                                                                                           \
        if (rc_ != 0) { srx_set_error(std::string(#call) + ": " + g_rccl.GetErrorString(rc_)); return 1; }          \
    } while (0)

// rows of `part`: groups of `group` rows dealt round-robin over nparts callers (srx_trace_pass_interleaved)
__host__ __device__ inline int rows_of_part(int res_y, int group, int nparts, int part) {
    int n = 0;
    for (int g0 = part * group; g0 < res_y; g0 += nparts * group) n += (res_y - g0 < group) ? (res_y - g0) : group;
    return n;
}
__host__ inline int most_rows(int res_y, int group, int nparts) { return rows_of_part(res_y, group, nparts, 0); }   // part 0 has the most

// slot layout: [buffer][local row][x] of fpp floats; local row j of `part` is image row ((j / group) * nparts + part) * group + j % group
__global__ __launch_bounds__(256) void k_rows_pack(const float* src, int fpp, int res_x, int group, int nparts, int part, int own_rows, float* slot) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_floats = (size_t)res_x * fpp;
    if (i >= (size_t)own_rows * row_floats) return;
    const int j = (int)(i / row_floats);
    const int y = ((j / group) * nparts + part) * group + (j % group);
    slot[i] = src[(size_t)y * row_floats + (i - (size_t)j * row_floats)];
}
// every part's slot (slot_stride floats apart) back into the frame; `skip_part` (the caller's own rows, already in place) is left out
__global__ __launch_bounds__(256) void k_rows_unpack(const float* slots, size_t slot_stride, size_t buffer_offset, int fpp, int res_x, int res_y, int group,
                                                    int nparts, int skip_part, float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_floats = (size_t)res_x * fpp;
    if (i >= (size_t)res_y * row_floats) return;
    const int y = (int)(i / row_floats);
    const int g = y / group, part = g % nparts;
    if (part == skip_part) return;
    const int j = (g / nparts) * group + (y % group);
    dst[i] = slots[(size_t)part * slot_stride + buffer_offset + (size_t)j * row_floats + (i - (size_t)y * row_floats)];
}

int check_rows(const char* who, int fpp, int res_x, int res_y, int group, int part, int nparts) {
    if (fpp <= 0 || fpp > 16 || res_x <= 0 || res_y <= 0) SRX_FAIL(std::string(who) + ": bad frame (floats per pixel, res_x, res_y)");
    if (group <= 0 || nparts <= 0 || part < -1 || part >= nparts) SRX_FAIL(std::string(who) + ": bad row interleave (group_rows, part, nparts)");
    return 0;
}

}  // namespace

struct srx_group {
    rccl_comm comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;
    float* d_slots = nullptr;          // nranks slots, grown on demand
    size_t slots_floats = 0;
};

extern "C" uint64_t srx_rows_slot_floats(int32_t floats_per_pixel, int32_t res_x, int32_t res_y, int32_t group_rows, int32_t nparts) {
    if (floats_per_pixel <= 0 || res_x <= 0 || res_y <= 0 || group_rows <= 0 || nparts <= 0) return 0;
    return (uint64_t)most_rows(res_y, group_rows, nparts) * (uint64_t)res_x * (uint64_t)floats_per_pixel;
}

extern "C" int srx_rows_pack(const float* d_frame, int32_t floats_per_pixel, int32_t res_x, int32_t res_y, int32_t group_rows, int32_t part,
                             int32_t nparts, float* d_slot, void* stream) {
    if (!d_frame || !d_slot) SRX_FAIL("srx_rows_pack: null argument");
    if (check_rows("srx_rows_pack", floats_per_pixel, res_x, res_y, group_rows, part, nparts) || part < 0) { if (part < 0) srx_set_error("srx_rows_pack: bad part"); return 1; }
    const int own = rows_of_part(res_y, group_rows, nparts, part);
    const size_t n = (size_t)own * res_x * floats_per_pixel;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_rows_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_frame, floats_per_pixel, res_x, group_rows, nparts,
                       part, own, d_slot);
    SRX_HIP(hipGetLastError());
    return 0;
}

extern "C" int srx_rows_unpack(const float* d_slots, uint64_t slot_stride_floats, uint64_t buffer_offset_floats, int32_t floats_per_pixel, int32_t res_x,
                               int32_t res_y, int32_t group_rows, int32_t nparts, int32_t skip_part, float* d_frame, void* stream) {
    if (!d_slots || !d_frame) SRX_FAIL("srx_rows_unpack: null argument");
    if (check_rows("srx_rows_unpack", floats_per_pixel, res_x, res_y, group_rows, skip_part, nparts)) return 1;
    const size_t n = (size_t)res_y * res_x * floats_per_pixel;
    hipLaunchKernelGGL(k_rows_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_slots, (size_t)slot_stride_floats,
                       (size_t)buffer_offset_floats, floats_per_pixel, res_x, res_y, group_rows, nparts, skip_part, d_frame);
    SRX_HIP(hipGetLastError());
    return 0;
}

extern "C" int srx_group_unique_id(void* id128) {
    if (!id128) SRX_FAIL("srx_group_unique_id: null argument");
    if (load_rccl()) return 1;
    rccl_unique_id id;
    SRX_RCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int srx_group_create(const void* id128, int32_t rank, int32_t nranks, srx_group** out) {
    if (!id128 || !out) SRX_FAIL("srx_group_create: null argument");
    if (nranks <= 0 || rank < 0 || rank >= nranks) SRX_FAIL("srx_group_create: bad rank / nranks");
    if (load_rccl()) return 1;
    srx_group* g = new srx_group();
    g->rank = rank; g->nranks = nranks;
    hipError_t e = hipGetDevice(&g->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete g; SRX_FAIL(std::string("srx_group_create: ") + hipGetErrorString(e)); }
    rccl_unique_id id;
    std::memcpy(&id, id128, sizeof(id));
    const int rc = g_rccl.CommInitRank(&g->comm, nranks, id, rank);
    if (rc != 0) { (void)hipStreamDestroy(g->stream); delete g; SRX_FAIL(std::string("srx_group_create: ncclCommInitRank: ") + g_rccl.GetErrorString(rc)); }
    *out = g;
    return 0;
}

extern "C" void srx_group_destroy(srx_group* g) {
    if (!g) return;
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->comm) (void)g_rccl.CommDestroy(g->comm);
    (void)hipFree(g->d_slots);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

namespace {
// A group belongs to the GPU that was current when it was made (its stream, its slots, its communicator): a collective
// called with another device current would launch the pack kernels and the RCCL calls from the wrong context.
int check_device(const srx_group* g, const char* who) {
    int dev = -1;
    SRX_HIP(hipGetDevice(&dev));
    if (dev != g->device)
        SRX_FAIL(std::string(who) + ": the group was made on device " + std::to_string(g->device) + " but device " + std::to_string(dev) +
                 " is current (hipSetDevice to the group's GPU before the call)");
    return 0;
}

int gather_rows(srx_group* g, float* const* frames, int n_frames, int fpp, int res_x, int res_y, int group_rows) {
    if (check_device(g, "srx_group_gather")) return 1;
    if (check_rows("srx_group_gather", fpp, res_x, res_y, group_rows, g->rank, g->nranks)) return 1;
    const size_t per_frame = (size_t)most_rows(res_y, group_rows, g->nranks) * res_x * fpp;
    const size_t slot = per_frame * n_frames;
    if (g->slots_floats < slot * g->nranks) {
        if (g->d_slots) { SRX_HIP(hipStreamSynchronize(g->stream)); SRX_HIP(hipFree(g->d_slots)); g->d_slots = nullptr; g->slots_floats = 0; }
        SRX_HIP(hipMalloc((void**)&g->d_slots, slot * g->nranks * sizeof(float)));
        g->slots_floats = slot * g->nranks;
    }
    // this rank's rows of every frame into its own slot (the all-gather is in place), one collective, then every other rank's rows back out
    float* mine = g->d_slots + (size_t)g->rank * slot;
    for (int f = 0; f < n_frames; f++)
        if (srx_rows_pack(frames[f], fpp, res_x, res_y, group_rows, g->rank, g->nranks, mine + (size_t)f * per_frame, g->stream)) return 1;
    SRX_RCCL(g_rccl.AllGather(mine, g->d_slots, slot, RCCL_FLOAT32, g->comm, g->stream));
    for (int f = 0; f < n_frames; f++)
        if (srx_rows_unpack(g->d_slots, slot, (size_t)f * per_frame, fpp, res_x, res_y, group_rows, g->nranks, g->rank, frames[f], g->stream)) return 1;
    SRX_HIP(hipStreamSynchronize(g->stream));
    return 0;
}
}  // namespace

extern "C" int srx_group_gather_framebuffer(srx_group* g, float* d_accum4, float* d_aux4, int32_t res_x, int32_t res_y, int32_t group_rows) {
    if (!g || !d_accum4) SRX_FAIL("srx_group_gather_framebuffer: null argument");
    float* frames[2] = {d_accum4, d_aux4};
    return gather_rows(g, frames, d_aux4 ? 2 : 1, 4, res_x, res_y, group_rows);
}

extern "C" int srx_group_allgather_rows(srx_group* g, float* d_frame, int32_t floats_per_pixel, int32_t res_x, int32_t res_y, int32_t group_rows) {
    if (!g || !d_frame) SRX_FAIL("srx_group_allgather_rows: null argument");
    float* frames[1] = {d_frame};
    return gather_rows(g, frames, 1, floats_per_pixel, res_x, res_y, group_rows);
}

namespace {
int gather_rows_dealt(srx_group* g, srx_deal* deal, float* const* frames, int n_frames, int fpp) {
    if (check_device(g, "srx_group_gather (dealt)")) return 1;
    if (deal->nparts != g->nranks) SRX_FAIL("srx_group_gather (dealt): the deal has " + std::to_string(deal->nparts) + " parts, the group " + std::to_string(g->nranks) + " ranks");
    if (fpp <= 0 || fpp > 16) SRX_FAIL("srx_group_gather (dealt): bad floats per pixel");
    const size_t per_frame = (size_t)srx_deal_most_rows(deal) * deal->res_x * fpp;
    const size_t slot = per_frame * n_frames;
    if (g->slots_floats < slot * g->nranks) {
        if (g->d_slots) { SRX_HIP(hipStreamSynchronize(g->stream)); SRX_HIP(hipFree(g->d_slots)); g->d_slots = nullptr; g->slots_floats = 0; }
        SRX_HIP(hipMalloc((void**)&g->d_slots, slot * g->nranks * sizeof(float)));
        g->slots_floats = slot * g->nranks;
    }
    float* mine = g->d_slots + (size_t)g->rank * slot;
    for (int f = 0; f < n_frames; f++)
        if (srx_rows_pack_dealt(deal, frames[f], fpp, g->rank, mine + (size_t)f * per_frame, g->stream)) return 1;
    SRX_RCCL(g_rccl.AllGather(mine, g->d_slots, slot, RCCL_FLOAT32, g->comm, g->stream));
    for (int f = 0; f < n_frames; f++)
        if (srx_rows_unpack_dealt(deal, g->d_slots, slot, (size_t)f * per_frame, fpp, g->rank, frames[f], g->stream)) return 1;
    SRX_HIP(hipStreamSynchronize(g->stream));
    return 0;
}
}  // namespace

extern "C" int srx_group_gather_framebuffer_dealt(srx_group* g, srx_deal* deal, float* d_accum4, float* d_aux4) {
    if (!g || !deal || !d_accum4) SRX_FAIL("srx_group_gather_framebuffer_dealt: null argument");
    float* frames[2] = {d_accum4, d_aux4};
    return gather_rows_dealt(g, deal, frames, d_aux4 ? 2 : 1, 4);
}

extern "C" int srx_group_allgather_rows_dealt(srx_group* g, srx_deal* deal, float* d_frame, int32_t floats_per_pixel) {
    if (!g || !deal || !d_frame) SRX_FAIL("srx_group_allgather_rows_dealt: null argument");
    float* frames[1] = {d_frame};
    return gather_rows_dealt(g, deal, frames, 1, floats_per_pixel);
}

extern "C" int srx_group_allreduce(srx_group* g, double* d_sums, uint32_t n_sums, int32_t* d_mins, uint32_t n_mins) {
    if (!g) SRX_FAIL("srx_group_allreduce: null group");
    if ((n_sums && !d_sums) || (n_mins && !d_mins)) SRX_FAIL("srx_group_allreduce: null argument");
    if (check_device(g, "srx_group_allreduce")) return 1;
    if (n_sums) SRX_RCCL(g_rccl.AllReduce(d_sums, d_sums, n_sums, RCCL_FLOAT64, RCCL_SUM, g->comm, g->stream));
    if (n_mins) SRX_RCCL(g_rccl.AllReduce(d_mins, d_mins, n_mins, RCCL_INT32, RCCL_MIN, g->comm, g->stream));
    SRX_HIP(hipStreamSynchronize(g->stream));
    return 0;
}

extern "C" int srx_group_rank(const srx_group* g, int32_t* rank, int32_t* nranks) {
    if (!g) SRX_FAIL("srx_group_rank: null group");
    if (rank) *rank = g->rank;
    if (nranks) *nranks = g->nranks;
    return 0;
}
