// raytracing_amd/csrc/srx_render.hip — C-ABI: wavefront renderer (host scheduler over the kernels).
//
// Replaces the loop around RayEngine::process_ray in RenderThread::process_iteratively
// (SceneManager.h:193-237) and the recursion inside process_ray (Core.h:592-598).
//
// Queue memory is one device array of `ray_capacity` 48-byte ray slots used as a STACK of segments:
// a segment [begin,end) of generation g is traced and shaded in batches; the children of a batch are
// appended above the current top and processed (depth first) before the next batch, then popped. A
// batch is sized so that its worst-case fan-out (bounce_count[g] children per ray, Core.h:428) fits in
// the free slots, so nothing is ever dropped and small capacities only mean more, smaller launches.
// Per batch: trace kernel -> shade kernel (appends children + shadow rays, one atomic per wave) ->
// count readback -> shadow kernel (adds lit direct-light contributions to the sample buffer).
#include "srx_host.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

#define SRX_RING 40           // chain mode: generations in flight without a host round trip (max_generations < 32)
#define SRX_COUNT_PAD 512     // dwords between the two queue counters (2 KB: measured, shade kernel 41.2 -> 40.2 ms per 4 passes)
namespace {

enum { CTR_NODES = 0, CTR_PACKETS = 1, CTR_SNODES = 2, CTR_SPACKETS = 3, CTR_HITS = 4, CTR_TEXELS = 5, CTR_PDF = 6,
       CTR_CHILDREN = 7, CTR_COUNT = 8, CTR_SLOTS = 256 };

// The two queue counters take a returning atomic per block / flush of the shade kernel (~0.9 M per pass): each on a
// memory channel of its own (the counters of the three sets are 4 KB apart as well).
struct DeviceCounts { uint32_t out_count; uint32_t pad0[SRX_COUNT_PAD - 1]; uint32_t shadow_count; uint32_t pad1[SRX_COUNT_PAD - 1]; };

}  // namespace

struct srx_renderer {
    srx_scene* scene = nullptr;
    uint64_t capacity = 0;            // ray slots
    uint64_t shadow_capacity = 0;
    GRay* rays = nullptr;
    GHit* hits = nullptr;
    // Shadow rays of a batch are traced on a second stream while the next generation's closest-hit kernel runs on the
    // first (they only share the atomically updated sample buffer and counters): two queues, used alternately, so the
    // shade kernel never writes the queue a shadow kernel is still reading.
    GShadow* shadow_q[2] = {nullptr, nullptr};
    hipStream_t shadow_stream = nullptr;
    hipEvent_t shadow_done[2] = {nullptr, nullptr};     // recorded on shadow_stream after the kernel that reads queue i
    bool shadow_busy[2] = {false, false};               // an un-joined shadow kernel uses queue i
    int shadow_flip = 0;
    bool overlap_shadow = true;
    // Chain mode (process_chain): generations whose fan-out is 1 are enqueued back to back, each kernel reading its ray
    // count from the counters of the one before (no host round trip per generation); the counts come back through a
    // pinned ring, one slot and one event per generation.
    bool chain = true;
    DeviceCounts* d_ring = nullptr;
    DeviceCounts* h_ring = nullptr;
    hipEvent_t ring_ev[SRX_RING];
    // kernel timing (set_timing): event pairs from a pool, resolved when the call ends
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Timed { int kind; size_t e0, e1; };
    std::vector<Timed> timed;
    DeviceCounts* d_counts = nullptr; // [3]: one per shadow queue, [2] for ray generation
    unsigned long long* d_counters = nullptr;
    float* sample3 = nullptr;         // per-pixel sample sums
    uint64_t sample_pixels = 0;
    bool sample_dirty = false;        // a call ended half way: its sums are still in sample3
    hipStream_t stream = nullptr;
    DeviceCounts* h_counts = nullptr; // pinned: [0] a batch's counts, [1].shadow_count the shadow count after the first round
    hipEvent_t round1_done = nullptr;
    bool shadow_early = false;        // launch_batch has launched the batch's shadow kernel itself (beside the second round)
    bool timing = false;
    // Rounds of the shade kernel (generations with a fan-out of 2 or more): the bounce loop of a ray runs until its slots are
    // used, attempts that are rejected do not count, and a wave runs as long as its slowest lane; a wave down to `spill_below`
    // busy lanes parks them (GCont) and the next round packs the parked loops of the whole batch 64 to a wave again.
    int shade_rounds = 2;             // SRX_SHADE_ROUNDS=1: one launch per batch, every wave to its end (3: park a second time)
    int spill_below = 16;
    int rounds_min_fan = 2;           // generations with at least this fan-out are shaded in rounds
    int park_pct = 100, park_add = 0; // a wave parks nothing before it has made bounce_count x park_pct / 100 + park_add trips (measured: 100 / 125 / 150 / 200 %)
    GCont* cont[2] = {nullptr, nullptr};
    uint64_t cont_cap[2] = {0, 0};
    uint32_t* d_cont_count = nullptr; // two counters, SRX_COUNT_PAD dwords apart (a third: the hit list's)
    uint32_t* h_cont_count = nullptr; // pinned [2]: what the last batch parked, checked against the arrays' sizes
    int last_rounds = 1;              // of the last launch_batch
    uint64_t last_need[2] = {0, 0};
    // Hit list: in an open scene many rays of the later generations leave the scene, and their lanes sit idle all through
    // the shade kernel. A generation whose batches hit less than `list_below` of the time (what the last batch of that
    // generation did: shadow rays / rays, known to the host anyway) is shaded over the list of its hits instead.
    float hit_ratio[32];
    float list_below = 0.8f;          // SRX_SHADE_HITLIST=0: never; =1: always (tests)
    uint32_t* hit_list = nullptr;
    uint64_t hit_list_cap = 0;
    // per-call accumulators
    uint64_t top = 0;
    srx_counters acc;
};

extern "C" int srx_renderer_create(srx_scene* scene, uint64_t ray_capacity, srx_renderer** out) {
    if (!scene || !out) SRX_FAIL("srx_renderer_create: null argument");
    if (scene->g.n_packets > 0 && (scene->g.n_materials == 0 || scene->g.n_tris == 0))
        SRX_FAIL("srx_renderer_create: the scene has no materials or no shading triangles (it can only be cast against)");
    if (ray_capacity < 64) SRX_FAIL("srx_renderer_create: ray_capacity must be at least 64");
    if (ray_capacity > 0xFFFFFFF0ull) SRX_FAIL("srx_renderer_create: ray_capacity must fit 32 bits");
    srx_renderer* r = new srx_renderer();
    r->scene = scene;
    r->capacity = ray_capacity;
    r->shadow_capacity = ray_capacity;
    std::memset(&r->acc, 0, sizeof(r->acc));
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    chk(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    chk(hipMalloc((void**)&r->rays, ray_capacity * sizeof(GRay)));
    chk(hipMalloc((void**)&r->hits, ray_capacity * sizeof(GHit)));
    if (const char* e2 = getenv("SRX_SHADOW_STREAM")) r->overlap_shadow = atoi(e2) != 0;   // experiments: 0 = one stream
    {
        // The shadow kernels are off the critical path (the next generation waits for the shade kernel, not for them), so
        // their stream gets the device's lowest priority: the closest-hit kernel takes the wave slots it can use and the
        // shadow kernel fills what is left (atrium 1354 -> 1361, interior 1529 -> 1534 Mrays/s; with the HIGHEST priority
        // 1351 / 1529). SRX_SHADOW_PRIO=default|high for measurements.
        int lo = 0, hi = 0;
        const char* pe = getenv("SRX_SHADOW_PRIO");
        const bool plain = pe && pe[0] == 'd';
        if (!plain && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
            chk(hipStreamCreateWithPriority(&r->shadow_stream, hipStreamNonBlocking, (pe && pe[0] == 'h') ? hi : lo));
        else chk(hipStreamCreateWithFlags(&r->shadow_stream, hipStreamNonBlocking));
    }
    if (const char* e3 = getenv("SRX_CHAIN")) r->chain = atoi(e3) != 0;                    // experiments / tests: 0 = a host round trip per generation
    if (const char* e4 = getenv("SRX_SHADE_ROUNDS")) r->shade_rounds = std::max(1, std::min(3, atoi(e4)));
    if (const char* e5 = getenv("SRX_SHADE_SPILL_BELOW")) r->spill_below = std::max(1, std::min(63, atoi(e5)));
    if (const char* e6 = getenv("SRX_SHADE_ROUNDS_MIN_FAN")) r->rounds_min_fan = std::max(2, atoi(e6));
    if (const char* e7 = getenv("SRX_SHADE_PARK_PCT")) r->park_pct = std::max(100, atoi(e7));
    if (const char* e8 = getenv("SRX_SHADE_PARK_ADD")) r->park_add = std::max(0, atoi(e8));
    chk(hipMalloc((void**)&r->d_cont_count, 3 * SRX_COUNT_PAD * sizeof(uint32_t)));
    for (int i = 0; i < 32; i++) r->hit_ratio[i] = 1.0f;
    if (const char* e9 = getenv("SRX_SHADE_HITLIST")) { const float v = (float)atof(e9); r->list_below = v >= 1.0f ? 2.0f : v; }
    for (int i = 0; i < 2; i++) {
        chk(hipMalloc((void**)&r->shadow_q[i], r->shadow_capacity * sizeof(GShadow)));
        chk(hipEventCreateWithFlags(&r->shadow_done[i], hipEventDisableTiming));
    }
    for (int i = 0; i < SRX_RING; i++) r->ring_ev[i] = nullptr;
    for (int i = 0; i < SRX_RING; i++) chk(hipEventCreateWithFlags(&r->ring_ev[i], hipEventDisableTiming));
    chk(hipMalloc((void**)&r->d_ring, SRX_RING * sizeof(DeviceCounts)));
    chk(hipHostMalloc((void**)&r->h_ring, SRX_RING * sizeof(DeviceCounts)));
    chk(hipMalloc((void**)&r->d_counts, 3 * sizeof(DeviceCounts)));
    chk(hipMalloc((void**)&r->d_counters, (size_t)CTR_SLOTS * CTR_COUNT * sizeof(unsigned long long)));
    chk(hipHostMalloc((void**)&r->h_counts, 2 * sizeof(DeviceCounts)));
    chk(hipEventCreateWithFlags(&r->round1_done, hipEventDisableTiming));
    chk(hipHostMalloc((void**)&r->h_cont_count, 2 * sizeof(uint32_t)));
    if (e != hipSuccess) {
        srx_set_error(std::string("srx_renderer_create: ") + hipGetErrorString(e));
        srx_renderer_destroy(r);
        return 1;
    }
    *out = r;
    return 0;
}

extern "C" void srx_renderer_destroy(srx_renderer* r) {
    if (!r) return;
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    if (r->shadow_stream) (void)hipStreamSynchronize(r->shadow_stream);
    (void)hipFree(r->rays); (void)hipFree(r->hits);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(r->shadow_q[i]);
        if (r->shadow_done[i]) (void)hipEventDestroy(r->shadow_done[i]);
    }
    for (int i = 0; i < SRX_RING; i++) if (r->ring_ev[i]) (void)hipEventDestroy(r->ring_ev[i]);
    for (hipEvent_t e : r->ev_pool) (void)hipEventDestroy(e);
    (void)hipFree(r->d_ring);
    if (r->h_ring) (void)hipHostFree(r->h_ring);
    if (r->shadow_stream) (void)hipStreamDestroy(r->shadow_stream);
    (void)hipFree(r->d_counts); (void)hipFree(r->d_counters); (void)hipFree(r->sample3);
    (void)hipFree(r->cont[0]); (void)hipFree(r->cont[1]); (void)hipFree(r->d_cont_count); (void)hipFree(r->hit_list);
    if (r->h_counts) (void)hipHostFree(r->h_counts);
    if (r->round1_done) (void)hipEventDestroy(r->round1_done);
    if (r->h_cont_count) (void)hipHostFree(r->h_cont_count);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

extern "C" int srx_renderer_set_timing(srx_renderer* r, int enabled) {
    if (!r) SRX_FAIL("srx_renderer_set_timing: null renderer");
    r->timing = enabled != 0;
    return 0;
}

namespace {

int ensure_sample_buffer(srx_renderer* r, uint64_t pixels) {
    if (r->sample_pixels >= pixels) return 0;
    if (r->sample3) { SRX_HIP(hipStreamSynchronize(r->stream)); SRX_HIP(hipFree(r->sample3)); r->sample3 = nullptr; }
    SRX_HIP(hipMalloc((void**)&r->sample3, pixels * 3 * sizeof(float)));
    SRX_HIP(hipMemsetAsync(r->sample3, 0, pixels * 3 * sizeof(float), r->stream));
    r->sample_pixels = pixels;
    return 0;
}

// ---- kernel timing: an event pair per timed kernel, taken from a pool and read when the call ends (nothing waits on
// the host for a time while work is being enqueued)
enum { T_TRACE = 0, T_SHADE = 1, T_SHADOW = 2 };
int t_begin(srx_renderer* r, int kind, hipStream_t st, size_t* slot) {
    *slot = (size_t)-1;
    if (!r->timing) return 0;
    while (r->ev_used + 2 > r->ev_pool.size()) { hipEvent_t e; SRX_HIP(hipEventCreate(&e)); r->ev_pool.push_back(e); }
    srx_renderer::Timed t{kind, r->ev_used, r->ev_used + 1};
    r->ev_used += 2;
    SRX_HIP(hipEventRecord(r->ev_pool[t.e0], st));
    r->timed.push_back(t);
    *slot = r->timed.size() - 1;
    return 0;
}
int t_end(srx_renderer* r, size_t slot, hipStream_t st) {
    if (slot == (size_t)-1) return 0;
    SRX_HIP(hipEventRecord(r->ev_pool[r->timed[slot].e1], st));
    return 0;
}
int resolve_timing(srx_renderer* r) {       // both streams have been waited for
    for (const srx_renderer::Timed& t : r->timed) {
        float ms = 0;
        SRX_HIP(hipEventElapsedTime(&ms, r->ev_pool[t.e0], r->ev_pool[t.e1]));
        if (t.kind == T_TRACE) r->acc.trace_ms += ms; else if (t.kind == T_SHADE) r->acc.shade_ms += ms; else r->acc.shadow_ms += ms;
    }
    r->timed.clear(); r->ev_used = 0;
    return 0;
}

// The shadow kernel that read queue i has to be over before the queue is written again / before the sample buffer is
// read: either the host waits for it, or the main stream does on the device.
int retire_shadow(srx_renderer* r, int i, bool host_wait) {
    if (!r->shadow_busy[i]) return 0;
    if (host_wait) {
        SRX_HIP(hipEventSynchronize(r->shadow_done[i]));
        r->shadow_busy[i] = false;
    } else {
        SRX_HIP(hipStreamWaitEvent(r->stream, r->shadow_done[i], 0));
    }
    return 0;
}

// every shadow kernel of this call is ordered before whatever the main stream does next (reads of the sample buffer)
int join_shadows(srx_renderer* r) {
    for (int i = 0; i < 2; i++) if (retire_shadow(r, i, false)) return 1;
    return 0;
}

// The shadow rays of a batch whose number the host has: on the second stream, beside whatever the main stream does next.
// The shade launch that filled queue q is over (the host has waited for it). Lit rays add their contribution to the sample buffer.
int launch_shadow_counted(srx_renderer* r, int q, uint64_t n_shadow) {
    if (!n_shadow) return 0;
    const GScene& g = r->scene->g;
    TraceArgs sh;
    std::memset(&sh, 0, sizeof(sh));
    sh.org = &r->shadow_q[q][0].ox; sh.org_stride = sizeof(GShadow) / 4;
    sh.n = (uint32_t)n_shadow;
    sh.shadow_q = r->shadow_q[q]; sh.sample3 = r->sample3;
    sh.counters = r->d_counters + CTR_SNODES; sh.counter_slot_mask = CTR_SLOTS - 1; sh.counter_slot_stride = CTR_COUNT;
    hipStream_t ss = r->overlap_shadow ? r->shadow_stream : r->stream;
    size_t ts;
    if (t_begin(r, T_SHADOW, ss, &ts)) return 1;
    SRX_HIP(srx_launch_trace(2, g, sh, ss));
    if (t_end(r, ts, ss)) return 1;
    SRX_HIP(hipEventRecord(r->shadow_done[q], ss));
    r->shadow_busy[q] = true;
    r->shadow_flip ^= 1;
    r->acc.shadow_launches++;
    return 0;
}

// One batch of rays through the closest-hit kernel, the shade kernel and (on the second stream) the shadow kernel.
// n_ptr (device, optional): the batch holds min(*n_ptr, n) rays, n only sizes the grids. The children go to `out`, their
// number and the number of shadow rays to `dc` (device). q = the shadow queue the shade kernel fills.
int launch_batch(srx_renderer* r, int gen, const GRay* in, uint32_t n, const uint32_t* n_ptr, GRay* out, DeviceCounts* dc, int q,
                 bool shadow_count_on_device, uint32_t n_hint, hipEvent_t shade_done) {
    const GScene& g = r->scene->g;
    size_t ts;
    TraceArgs t;
    std::memset(&t, 0, sizeof(t));
    t.org = &in->ox; t.org_stride = sizeof(GRay) / 4;
    t.dir = &in->dx; t.dir_stride = sizeof(GRay) / 4;
    t.n = n; t.n_ptr = n_ptr; t.n_hint = n_hint;
    t.hits = r->hits;
    t.counters = r->d_counters + CTR_NODES; t.counter_slot_mask = CTR_SLOTS - 1; t.counter_slot_stride = CTR_COUNT;
    if (t_begin(r, T_TRACE, r->stream, &ts)) return 1;
    SRX_HIP(srx_launch_trace(0, g, t, r->stream));
    if (t_end(r, ts, r->stream)) return 1;

    ShadeArgs s;
    std::memset(&s, 0, sizeof(s));
    s.rays = in; s.hits = r->hits; s.n = n; s.n_ptr = n_ptr; s.generation = gen;
    s.out_rays = out; s.out_count = &dc->out_count;
    s.shadow_q = r->shadow_q[q]; s.shadow_count = &dc->shadow_count;
    s.sample3 = r->sample3; s.counters = r->d_counters; s.counter_slot_mask = CTR_SLOTS - 1; s.counter_slot_stride = CTR_COUNT;
    // rounds: for the batches whose size the host knows (fan-out of 2 or more: the synchronous form)
    const int rounds = (!n_ptr && gen < g.max_generations && g.bounce_count[gen] >= r->rounds_min_fan) ? r->shade_rounds : 1;
    uint64_t need[2] = {0, 0};
    if (rounds > 1) {
        need[0] = ((uint64_t)n + 63) / 64 * (uint64_t)r->spill_below;          // a wave parks spill_below lanes at most
        need[1] = (need[0] + 63) / 64 * (uint64_t)r->spill_below;
        for (int k = 0; k < rounds - 1; k++) {
            if (r->cont_cap[k] >= need[k]) continue;
            SRX_HIP(hipStreamSynchronize(r->stream));                             // nothing enqueued may still use the old array
            if (r->cont[k]) { SRX_HIP(hipFree(r->cont[k])); r->cont[k] = nullptr; r->cont_cap[k] = 0; }
            const uint64_t cap = need[k] + need[k] / 4 + 1024;                    // batches of a generation differ a little from pass to pass: room to spare, so that the array is not made again every time one is a few rays larger
            SRX_HIP(hipMalloc((void**)&r->cont[k], cap * sizeof(GCont)));
            r->cont_cap[k] = cap;
        }
        SRX_HIP(hipMemsetAsync(r->d_cont_count, 0, 2 * SRX_COUNT_PAD * sizeof(uint32_t), r->stream));
    }
    // the hit list, for a generation whose rays mostly miss (sun-less scenes have no shadow count to tell by: never)
    const bool listed = g.n_suns > 0 && gen < 32 && r->hit_ratio[gen] < r->list_below;
    if (listed) {
        if (r->hit_list_cap < n) {
            SRX_HIP(hipStreamSynchronize(r->stream));
            if (r->hit_list) { SRX_HIP(hipFree(r->hit_list)); r->hit_list = nullptr; r->hit_list_cap = 0; }
            const uint64_t cap = (uint64_t)n + n / 4 + 1024;
            SRX_HIP(hipMalloc((void**)&r->hit_list, cap * sizeof(uint32_t)));
            r->hit_list_cap = cap;
        }
        uint32_t* const lc = r->d_cont_count + 2 * SRX_COUNT_PAD;
        SRX_HIP(hipMemsetAsync(lc, 0, sizeof(uint32_t), r->stream));
        s.slot_list = r->hit_list; s.n_ptr = lc;
    }
    if (t_begin(r, T_SHADE, r->stream, &ts)) return 1;
    if (listed) SRX_HIP(srx_launch_hitlist(r->hits, n, n_ptr, r->hit_list, r->d_cont_count + 2 * SRX_COUNT_PAD, r->stream));
    for (int k = 0; k < rounds; k++) {
        ShadeArgs sk = s;
        if (k > 0) { sk.cont_in = r->cont[k - 1]; sk.n = (uint32_t)need[k - 1]; sk.n_ptr = r->d_cont_count + (size_t)(k - 1) * SRX_COUNT_PAD; }
        if (k + 1 < rounds) {
            sk.cont_out = r->cont[k]; sk.cont_count = r->d_cont_count + (size_t)k * SRX_COUNT_PAD; sk.cont_capacity = (uint32_t)need[k];
            sk.spill_below = r->spill_below;
            sk.spill_after = std::max(1, g.bounce_count[gen]) * r->park_pct / 100 + r->park_add;
        }
        SRX_HIP(srx_launch_shade(g, sk, r->stream));
        r->acc.shade_launches++;
        // every shadow ray of the batch comes from the first round (a parked loop has queued its ray's already): with two
        // streams their kernel starts as soon as that round is over, beside the later rounds, whose few waves with the longest loops leave most of the chip idle
        if (k == 0 && rounds > 1 && !shadow_count_on_device && g.n_suns && r->overlap_shadow) {
            SRX_HIP(hipMemcpyAsync(&r->h_counts[1].shadow_count, &dc->shadow_count, sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
            SRX_HIP(hipEventRecord(r->round1_done, r->stream));
            r->shadow_early = true;
        }
    }
    if (t_end(r, ts, r->stream)) return 1;
    r->last_rounds = rounds; r->last_need[0] = need[0]; r->last_need[1] = need[1];
    if (r->shadow_early) {
        SRX_HIP(hipEventSynchronize(r->round1_done));
        const uint64_t n_shadow = r->h_counts[1].shadow_count;
        if (n_shadow > (uint64_t)n * g.n_suns) SRX_FAIL("srx: internal error, shadow count exceeds the queue");
        if (launch_shadow_counted(r, q, n_shadow)) return 1;
    }
    r->acc.trace_launches++;
    static const bool rounds_log = getenv("SRX_ROUNDS_LOG") != nullptr;
    if (rounds_log && rounds > 1) {
        uint32_t c[2] = {0, 0};
        SRX_HIP(hipStreamSynchronize(r->stream));
        SRX_HIP(hipMemcpy(&c[0], r->d_cont_count, 4, hipMemcpyDeviceToHost));
        SRX_HIP(hipMemcpy(&c[1], r->d_cont_count + SRX_COUNT_PAD, 4, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "srx rounds gen %d: %u rays, parked %u then %u\n", gen, n, c[0], c[1]);
    }
    if (!shadow_count_on_device) return 0;          // the caller launches the shadow kernel once it has the count
    if (g.n_suns) {
        // shadow rays of this batch, counted on the device: grid for the most there can be
        SRX_HIP(hipEventRecord(shade_done, r->stream));
        hipStream_t ss = r->overlap_shadow ? r->shadow_stream : r->stream;
        if (ss != r->stream) SRX_HIP(hipStreamWaitEvent(ss, shade_done, 0));
        TraceArgs sh;
        std::memset(&sh, 0, sizeof(sh));
        sh.org = &r->shadow_q[q][0].ox; sh.org_stride = sizeof(GShadow) / 4;
        sh.n = n * g.n_suns; sh.n_ptr = &dc->shadow_count; sh.n_hint = n_hint * g.n_suns;
        sh.shadow_q = r->shadow_q[q]; sh.sample3 = r->sample3;
        sh.counters = r->d_counters + CTR_SNODES; sh.counter_slot_mask = CTR_SLOTS - 1; sh.counter_slot_stride = CTR_COUNT;
        if (t_begin(r, T_SHADOW, ss, &ts)) return 1;
        SRX_HIP(srx_launch_trace(2, g, sh, ss));
        if (t_end(r, ts, ss)) return 1;
        SRX_HIP(hipEventRecord(r->shadow_done[q], ss));
        r->shadow_busy[q] = true;
        r->shadow_flip ^= 1;
        r->acc.shadow_launches++;
    }
    return 0;
}

// Chain mode: the batch [begin, begin + n) of generation `gen` and ALL its descendants, for generations whose fan-out
// is 1 (bounce_count <= 1: from the fourth generation on with the reference's 32 x 0.25^g, every generation of a pure
// path tracer). A ray then has at most one child, so n slots hold any generation: the generations ping-pong between two
// regions above the top and every kernel takes its ray count from the counters the previous shade kernel left on the
// device. The device never waits for the host: the counts come back through a pinned ring, and the host, two links
// ahead at most, uses them to tighten the grids of the links still to be enqueued and to end the chain once a generation is empty.
int process_chain(srx_renderer* r, int gen, uint64_t begin, uint64_t n) {
    const GScene& g = r->scene->g;
    const int links = g.max_generations - gen + 1;
    GRay* const region[2] = {r->rays + r->top, r->rays + r->top + n};
    // The shadow kernels of an earlier chain of this call (second stream, lowest priority) read their ray counts from the
    // ring when their blocks START: the ring is cleared only after every one of them is over, on the device (the main
    // stream waits for both queues' events; the host does not). Without this a block dispatched after the memset saw a
    // count of 0 and dropped its shadow rays (tests/test_shade_gpu.py::test_chains_back_to_back).
    if (join_shadows(r)) return 1;
    SRX_HIP(hipMemsetAsync(r->d_ring, 0, (size_t)links * sizeof(DeviceCounts), r->stream));
    const GRay* in = r->rays + begin;
    const uint32_t* n_ptr = nullptr;
    uint32_t bound = (uint32_t)n;
    // What the host expects link k to hold (it knows the counts up to link k - 2): the last known count shrunk by the
    // last known ratio once per link since. Only the form of the traversal kernel depends on it (rays per octet: a launch
    // that turns out small should not run a few long waves); the grid is sized for `bound`.
    double known = (double)n, ratio = 1.0;
    int known_at = 0;
    static const bool chain_log = getenv("SRX_CHAIN_LOG") != nullptr;
    int used = 0;
    for (int k = 0; k < links; k++) {
        const int q = r->shadow_flip;
        if (retire_shadow(r, q, false)) return 1;            // the shadow kernel that read queue q: the main stream waits on the device
        const uint32_t hint = (uint32_t)std::max(1.0, std::min((double)bound, known * std::pow(ratio, k - known_at)));
        if (launch_batch(r, gen + k, in, bound, n_ptr, region[k & 1], r->d_ring + k, q, true, hint, r->ring_ev[k])) return 1;
        SRX_HIP(hipMemcpyAsync(r->h_ring + k, r->d_ring + k, sizeof(DeviceCounts), hipMemcpyDeviceToHost, r->stream));
        // (launch_batch recorded the same event after the shade kernel for the shadow stream to wait on; that wait was
        // enqueued then and keeps referring to that record. This record, after the copy, is the one the host waits for.)
        SRX_HIP(hipEventRecord(r->ring_ev[k], r->stream));
        used = k + 1;
        if (gen + k >= g.max_generations) break;              // Core.h:473: this generation queued nothing
        // The host stays at most two links ahead of the device: before link k + 1 is enqueued it has the count of link
        // k - 1 (whose kernels are over or nearly so, while link k keeps the device busy), which bounds the rays of
        // every later link and ends the chain when it is 0.
        if (k >= 1) {
            SRX_HIP(hipEventSynchronize(r->ring_ev[k - 1]));
            const uint32_t c = r->h_ring[k - 1].out_count;    // rays of link k; link k + 1 has no more
            if (c == 0) break;
            bound = std::min(bound, c);
            ratio = std::min(1.0, (double)c / std::max(1.0, known));
            known = (double)c; known_at = k;
        }
        in = region[k & 1];
        n_ptr = &r->d_ring[k].out_count;
    }
    SRX_HIP(hipStreamSynchronize(r->stream));
    uint64_t rays = n;
    if (chain_log) {
        std::fprintf(stderr, "srx chain gen %d: %llu", gen, (unsigned long long)n);
        for (int k = 0; k < used; k++) std::fprintf(stderr, " -> %u (+%u shadow)", r->h_ring[k].out_count, r->h_ring[k].shadow_count);
        std::fprintf(stderr, "\n");
    }
    for (int k = 0; k < used; k++) {
        const uint64_t c = r->h_ring[k].out_count, sh = r->h_ring[k].shadow_count;
        if (c > n || sh > n * g.n_suns) SRX_FAIL("srx: internal error, a chained generation grew");
        const uint64_t held = k == 0 ? n : (uint64_t)r->h_ring[k - 1].out_count;      // rays of link k
        if (g.n_suns && held && gen + k < 32) r->hit_ratio[gen + k] = (float)((double)sh / ((double)held * g.n_suns));
        if (k + 1 < used) rays += c;
        else if (c != 0 && gen + k < g.max_generations) SRX_FAIL("srx: internal error, a chain ended with rays left");
        r->acc.shadows_cast += sh;
    }
    r->acc.rays_processed += rays;
    return 0;
}

// Process the queue segment [begin, end) of generation `gen`; children go above r->top.
int process_segment(srx_renderer* r, int gen, uint64_t begin, uint64_t end) {
    const GScene& g = r->scene->g;
    const bool spawns = gen < g.max_generations;                        // Core.h:473
    const uint64_t fan = spawns ? (uint64_t)std::max(1, g.bounce_count[gen]) : 0;
    bool chain_ok = r->chain && g.max_generations - gen + 1 <= SRX_RING;
    for (int d = gen; d < g.max_generations && chain_ok; d++) chain_ok = g.bounce_count[d] <= 1;
    while (begin < end) {
        uint64_t n = end - begin;
        if (fan) {
            // keep `reserve` slots back so every deeper generation can still process at least one ray at a time
            uint64_t reserve = 0;
            for (int d = gen + 1; d < g.max_generations; d++) reserve += (uint64_t)std::max(1, g.bounce_count[d]);
            const uint64_t free_slots = r->capacity - r->top;
            n = std::min(n, free_slots > reserve ? (free_slots - reserve) / fan : 0);
            if (n == 0) SRX_FAIL("srx: ray_capacity too small for one ray's fan-out at this generation; create the renderer with more slots");
        }
        if (g.n_suns) n = std::min(n, r->shadow_capacity / g.n_suns);
        if (n == 0) SRX_FAIL("srx: shadow queue too small");

        if (chain_ok && fan == 1 && r->top + 2 * n <= r->capacity) {
            if (process_chain(r, gen, begin, n)) return 1;
            begin += n;
            continue;
        }

        // ---- closest hit, shade + bounce (children appended at top, shadow rays into queue q), then the counts
        const int q = r->shadow_flip;
        if (retire_shadow(r, q, true)) return 1;       // the kernel that read queue q two batches ago (over long since)
        DeviceCounts* const dc = r->d_counts + q;
        SRX_HIP(hipMemsetAsync(dc, 0, sizeof(DeviceCounts), r->stream));
        if (launch_batch(r, gen, r->rays + begin, (uint32_t)n, nullptr, r->rays + r->top, dc, q, false, 0, nullptr)) return 1;
        SRX_HIP(hipMemcpyAsync(r->h_counts, dc, sizeof(DeviceCounts), hipMemcpyDeviceToHost, r->stream));
        for (int k = 0; k + 1 < r->last_rounds; k++)
            SRX_HIP(hipMemcpyAsync(r->h_cont_count + k, r->d_cont_count + (size_t)k * SRX_COUNT_PAD, sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
        SRX_HIP(hipStreamSynchronize(r->stream));
        // a wave parks spill_below loops at most, which is what the arrays were sized by: more would have been dropped
        for (int k = 0; k + 1 < r->last_rounds; k++)
            if (r->h_cont_count[k] > r->last_need[k]) SRX_FAIL("srx: internal error, more parked bounce loops than their array holds");
        const uint64_t n_children = r->h_counts->out_count, n_shadow = r->h_counts->shadow_count;
        if (n_children > n * fan || r->top + n_children > r->capacity) SRX_FAIL("srx: internal error, child count exceeds the reserved slots");
        if (n_shadow > n * g.n_suns) SRX_FAIL("srx: internal error, shadow count exceeds the queue");
        if (g.n_suns && gen < 32) r->hit_ratio[gen] = (float)((double)n_shadow / ((double)n * g.n_suns));

        // ---- shadow rays of this batch, on the second stream: the kernel runs next to the closest-hit kernel of the
        // next generation (launched below on the main stream). The shade kernel that filled the queue is over (host
        // synchronised above). Lit rays add their contribution to the sample buffer. (For these large batches, launching
        // it before the synchronisation with the count read on the device measured slower: 1250 against 1263 Mrays/s.)
        if (!r->shadow_early && launch_shadow_counted(r, q, n_shadow)) return 1;
        r->shadow_early = false;
        r->acc.rays_processed += n;
        r->acc.shadows_cast += n_shadow;
        // ---- next generation, depth first
        if (n_children) {
            const uint64_t old_top = r->top;
            r->top += n_children;
            if (process_segment(r, gen + 1, old_top, old_top + n_children)) return 1;
            r->top = old_top;
        }
        begin += n;
    }
    return 0;
}

int begin_call(srx_renderer* r) {
    for (int i = 0; i < 2; i++) if (retire_shadow(r, i, true)) return 1;   // left over from a call that failed half way
    std::memset(&r->acc, 0, sizeof(r->acc));
    r->timed.clear(); r->ev_used = 0;
    r->shadow_early = false;
    // everything a failed call may have left running is over now (both shadow queues retired above, and every error path
    // of this file comes after a synchronisation of the main stream or leaves nothing enqueued on it that uses sample3)
    SRX_HIP(hipStreamSynchronize(r->stream));
    if (r->sample_dirty && r->sample3) SRX_HIP(hipMemsetAsync(r->sample3, 0, r->sample_pixels * 3 * sizeof(float), r->stream));
    r->sample_dirty = true;           // until end_call
    SRX_HIP(hipMemsetAsync(r->d_counters, 0, (size_t)CTR_SLOTS * CTR_COUNT * sizeof(unsigned long long), r->stream));
    return 0;
}

int end_call(srx_renderer* r, srx_counters* out) {
    static thread_local std::vector<unsigned long long> all;
    all.resize((size_t)CTR_SLOTS * CTR_COUNT);
    if (join_shadows(r)) return 1;
    SRX_HIP(hipMemcpyAsync(all.data(), r->d_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, r->stream));
    SRX_HIP(hipStreamSynchronize(r->stream));
    for (int i = 0; i < 2; i++) if (retire_shadow(r, i, true)) return 1;
    if (resolve_timing(r)) return 1;
    unsigned long long c[CTR_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sl = 0; sl < CTR_SLOTS; sl++) for (int k = 0; k < CTR_COUNT; k++) c[k] += all[(size_t)sl * CTR_COUNT + k];
    r->acc.nodes = c[CTR_NODES]; r->acc.packets = c[CTR_PACKETS];
    r->acc.shadow_nodes = c[CTR_SNODES]; r->acc.shadow_packets = c[CTR_SPACKETS];
    r->acc.hits = c[CTR_HITS]; r->acc.texels = c[CTR_TEXELS]; r->acc.pdf_lookups = c[CTR_PDF];
    r->acc.diffuses_cast = c[CTR_CHILDREN];
    if (out) *out = r->acc;
    r->sample_dirty = false;          // the accumulate kernels and srx_process_rays leave sample3 zeroed
    return srx_scene_check_spill(r->scene);
}

}  // namespace

extern "C" int srx_trace_pass_interleaved(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                                          int32_t group_rows, int32_t part, int32_t nparts, uint32_t pass_index, uint64_t seed,
                                          float* d_accum4, float* d_aux4, srx_counters* counters) {
    return srx_trace_passes_interleaved(r, camera, x0, y0, w, h, group_rows, part, nparts, pass_index, 1, seed, d_accum4, d_aux4, counters);
}

extern "C" int srx_trace_passes_interleaved(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                                            int32_t group_rows, int32_t part, int32_t nparts, uint32_t pass_index, int32_t n_passes,
                                            uint64_t seed, float* d_accum4, float* d_aux4, srx_counters* counters) {
    if (!r || !camera || !d_accum4) SRX_FAIL("srx_trace_pass: null argument");
    if (n_passes <= 0) SRX_FAIL("srx_trace_passes_interleaved: n_passes must be at least 1");
    if (camera->res_x <= 0 || camera->res_y <= 0 || camera->subdiv <= 0) SRX_FAIL("srx_trace_pass: bad camera");
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > camera->res_x || y0 + h > camera->res_y)
        SRX_FAIL("srx_trace_pass: rectangle outside the camera resolution");
    if (group_rows <= 0 || nparts <= 0 || part < 0 || part >= nparts) SRX_FAIL("srx_trace_pass_interleaved: bad row interleave (group_rows, part, nparts)");
    const uint64_t npix = (uint64_t)camera->res_x * camera->res_y;
    if (npix * (uint64_t)n_passes > 0x00FFFFFFull) SRX_FAIL("srx_trace_pass: more than 2^24-1 pixels x passes in flight is not supported");
    if (begin_call(r)) return 1;
    if (ensure_sample_buffer(r, npix * (uint64_t)n_passes)) return 1;
    static_assert(sizeof(GCamera) == sizeof(srx_camera), "camera layout");
    GCamera cam;
    std::memcpy(&cam, camera, sizeof(cam));
    // rows of this caller: row groups part, part + nparts, ... of the rectangle (only the last group can be short)
    GRows rm;
    rm.y0 = y0; rm.group = group_rows; rm.nparts = nparts; rm.part = part;
    int own_rows = 0;
    for (int g0 = part * group_rows; g0 < h; g0 += nparts * group_rows) own_rows += std::min(group_rows, h - g0);
    if (own_rows == 0) return end_call(r, counters);
    // primaries are generated in row bands that fit the queue with their worst-case fan-out
    const GScene& g = r->scene->g;
    const uint64_t fan0 = (uint64_t)std::max(1, g.bounce_count[0]);
    uint64_t band_rays = r->capacity / (1 + (g.max_generations > 0 ? fan0 : 0));
    if (band_rays < (uint64_t)w * (uint64_t)n_passes) SRX_FAIL("srx_trace_pass: ray_capacity too small for one pixel row of every pass and its first bounce");
    const int band_rows = (int)std::min<uint64_t>((uint64_t)own_rows, band_rays / ((uint64_t)w * (uint64_t)n_passes));
    for (int j = 0; j < own_rows; j += band_rows) {
        const int rows = std::min(band_rows, own_rows - j);
        const uint64_t n1 = (uint64_t)rows * w;
        // the primaries of all the passes of this band share the launches of every generation
        for (int p = 0; p < n_passes; p++)
            SRX_HIP(srx_launch_raygen(cam, x0, w, rm, j, rows, pass_index + (uint32_t)p, seed, (uint32_t)((uint64_t)p * npix), r->rays + (uint64_t)p * n1,
                                      nullptr, nullptr, r->stream));
        r->top = n1 * (uint64_t)n_passes;
        if (process_segment(r, 0, 0, n1 * (uint64_t)n_passes)) return 1;
    }
    if (join_shadows(r)) return 1;
    SRX_HIP(srx_launch_accumulate(camera->res_x, x0, w, rm, own_rows, pass_index, n_passes, (uint32_t)npix, r->sample3, d_accum4, d_aux4, r->stream));
    return end_call(r, counters);
}

namespace {
// One visit over the rows rm describes (own_rows of them), columns [x0, x0 + w).
int trace_visit(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t w, const GRows& rm, int own_rows, uint32_t visit_index, uint64_t seed,
                const float* d_block_priority, float* d_accum4, float* d_aux4, srx_counters* counters) {
    const uint64_t npix = (uint64_t)camera->res_x * camera->res_y;
    if (npix > 0x00FFFFFFull) SRX_FAIL("srx_trace_pass_adaptive: more than 2^24-1 pixels is not supported");
    if (begin_call(r)) return 1;
    if (ensure_sample_buffer(r, npix)) return 1;
    GCamera cam;
    std::memcpy(&cam, camera, sizeof(cam));
    if (own_rows == 0) return end_call(r, counters);
    AdaptiveArgs a;
    a.block_priority = d_block_priority; a.accum4 = (const float4*)d_accum4; a.visit_index = visit_index; a.seed = seed;
    a.blocks_x = (uint32_t)((camera->res_x + 3) / 4);
    a.n_blocks = a.blocks_x * (uint32_t)((camera->res_y + 3) / 4);
    const GScene& g = r->scene->g;
    const uint64_t fan0 = (uint64_t)std::max(1, g.bounce_count[0]);
    uint64_t band_rays = r->capacity / (1 + (g.max_generations > 0 ? fan0 : 0));
    if (band_rays < (uint64_t)w) SRX_FAIL("srx_trace_pass_adaptive: ray_capacity too small for one pixel row and its first bounce");
    const int band_rows = (int)std::min<uint64_t>((uint64_t)own_rows, band_rays / (uint64_t)w);
    for (int j = 0; j < own_rows; j += band_rows) {
        const int rows = std::min(band_rows, own_rows - j);
        // primaries of the blocks that fire, appended compactly; their number comes back with an 8-byte copy
        SRX_HIP(hipMemsetAsync(r->d_counts + 2, 0, sizeof(DeviceCounts), r->stream));
        SRX_HIP(srx_launch_raygen_adaptive(cam, x0, w, rm, j, rows, a, r->rays, &r->d_counts[2].out_count, r->stream));
        SRX_HIP(hipMemcpyAsync(r->h_counts, r->d_counts + 2, sizeof(DeviceCounts), hipMemcpyDeviceToHost, r->stream));
        SRX_HIP(hipStreamSynchronize(r->stream));
        const uint64_t n = r->h_counts->out_count;
        if (n > (uint64_t)rows * w) SRX_FAIL("srx: internal error, more primaries than pixels");
        r->top = n;
        if (n && process_segment(r, 0, 0, n)) return 1;
    }
    if (join_shadows(r)) return 1;
    SRX_HIP(srx_launch_accumulate_adaptive(camera->res_x, x0, w, rm, own_rows, a, r->sample3, d_accum4, d_aux4, r->stream));
    return end_call(r, counters);
}
}  // namespace

extern "C" int srx_trace_pass_adaptive(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                                       int32_t group_rows, int32_t part, int32_t nparts, uint32_t visit_index, uint64_t seed,
                                       const float* d_block_priority, float* d_accum4, float* d_aux4, srx_counters* counters) {
    if (!r || !camera || !d_accum4) SRX_FAIL("srx_trace_pass_adaptive: null argument");
    if (camera->res_x <= 0 || camera->res_y <= 0 || camera->subdiv <= 0) SRX_FAIL("srx_trace_pass_adaptive: bad camera");
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > camera->res_x || y0 + h > camera->res_y)
        SRX_FAIL("srx_trace_pass_adaptive: rectangle outside the camera resolution");
    if (group_rows <= 0 || nparts <= 0 || part < 0 || part >= nparts) SRX_FAIL("srx_trace_pass_adaptive: bad row interleave (group_rows, part, nparts)");
    GRows rm;
    rm.y0 = y0; rm.group = group_rows; rm.nparts = nparts; rm.part = part;
    int own_rows = 0;
    for (int g0 = part * group_rows; g0 < h; g0 += nparts * group_rows) own_rows += std::min(group_rows, h - g0);
    return trace_visit(r, camera, x0, w, rm, own_rows, visit_index, seed, d_block_priority, d_accum4, d_aux4, counters);
}

// The same visit over the rows `part` owns under a deal (csrc/srx_deal.hip): the whole frame's rows, dealt by burden.
extern "C" int srx_trace_pass_adaptive_dealt(srx_renderer* r, const srx_camera* camera, srx_deal* deal, int32_t part, uint32_t visit_index, uint64_t seed,
                                             const float* d_block_priority, float* d_accum4, float* d_aux4, srx_counters* counters) {
    if (!r || !camera || !d_accum4 || !deal) SRX_FAIL("srx_trace_pass_adaptive_dealt: null argument");
    if (camera->res_x <= 0 || camera->res_y <= 0 || camera->subdiv <= 0) SRX_FAIL("srx_trace_pass_adaptive_dealt: bad camera");
    if (deal->res_x != camera->res_x || deal->res_y != camera->res_y) SRX_FAIL("srx_trace_pass_adaptive_dealt: the deal was made for another frame size");
    if (part < 0 || part >= deal->nparts) SRX_FAIL("srx_trace_pass_adaptive_dealt: part out of range");
    GRows rm;
    rm.y0 = 0; rm.group = deal->group; rm.nparts = deal->nparts; rm.part = part;
    if (srx_deal_device(deal, part, &rm.list, &rm.owner, nullptr)) return 1;
    return trace_visit(r, camera, 0, camera->res_x, rm, srx_deal_rows_of_part(deal, part), visit_index, seed, d_block_priority, d_accum4, d_aux4, counters);
}

extern "C" int srx_trace_pass(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                              uint32_t pass_index, uint64_t seed, float* d_accum4, float* d_aux4, srx_counters* counters) {
    return srx_trace_pass_interleaved(r, camera, x0, y0, w, h, h > 0 ? h : 1, 0, 1, pass_index, seed, d_accum4, d_aux4, counters);
}

extern "C" int srx_process_rays(srx_renderer* r, const float* org3, const float* dir3, const uint64_t* keys, int32_t generation,
                                uint32_t n, float* radiance3, srx_counters* counters) {
    if (!r || !org3 || !dir3 || !keys || !radiance3) SRX_FAIL("srx_process_rays: null argument");
    if (generation < 0 || generation >= 32) SRX_FAIL("srx_process_rays: generation out of range");
    if (n > 0x00FFFFFFu) SRX_FAIL("srx_process_rays: at most 2^24-1 rays per call");
    if (begin_call(r)) return 1;
    if (n == 0) return end_call(r, counters);
    if (ensure_sample_buffer(r, n)) return 1;
    const GScene& g = r->scene->g;
    const uint64_t fan = generation < g.max_generations ? (uint64_t)std::max(1, g.bounce_count[generation]) : 0;
    const uint64_t chunk = r->capacity / (1 + fan);
    if (chunk == 0) SRX_FAIL("srx_process_rays: ray_capacity too small");
    float *d_o = nullptr, *d_d = nullptr; unsigned long long* d_k = nullptr;
    struct Free { float*& a; float*& b; unsigned long long*& c; ~Free() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); } } free_all{d_o, d_d, d_k};
    SRX_HIP(hipMalloc((void**)&d_o, (size_t)n * 12)); SRX_HIP(hipMalloc((void**)&d_d, (size_t)n * 12)); SRX_HIP(hipMalloc((void**)&d_k, (size_t)n * 8));
    int rc = 0;
    do {
        if (hipMemcpy(d_o, org3, (size_t)n * 12, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_d, dir3, (size_t)n * 12, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_k, keys, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { srx_set_error("srx_process_rays: upload failed"); rc = 1; break; }
        for (uint64_t b = 0; b < n && !rc; b += chunk) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - b);
            if (srx_launch_pack_rays(d_o + 3 * b, d_d + 3 * b, d_k + b, m, r->rays, r->stream) != hipSuccess) { srx_set_error("srx_process_rays: pack launch failed"); rc = 1; break; }
            // pixel index of ray i is i within the chunk: shift by the chunk base afterwards
            r->top = m;
            if (process_segment(r, generation, 0, m) || join_shadows(r)) { rc = 1; break; }
            if (hipMemcpyAsync(radiance3 + 3 * b, r->sample3, (size_t)m * 12, hipMemcpyDeviceToHost, r->stream) != hipSuccess ||
                hipMemsetAsync(r->sample3, 0, (size_t)m * 12, r->stream) != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess) {
                srx_set_error("srx_process_rays: readback failed"); rc = 1; break;
            }
        }
    } while (0);
    if (rc) return rc;
    return end_call(r, counters);
}

extern "C" int srx_camera_rays(const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h, uint32_t pass_index,
                               uint64_t seed, float* dir3, uint64_t* keys) {
    if (!camera || !dir3 || !keys) SRX_FAIL("srx_camera_rays: null argument");
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > camera->res_x || y0 + h > camera->res_y)
        SRX_FAIL("srx_camera_rays: rectangle outside the camera resolution");
    GCamera cam;
    std::memcpy(&cam, camera, sizeof(cam));
    const size_t n = (size_t)w * h;
    float* d_d = nullptr; unsigned long long* d_k = nullptr;
    SRX_HIP(hipMalloc((void**)&d_d, n * 12)); SRX_HIP(hipMalloc((void**)&d_k, n * 8));
    GRows rm;
    rm.y0 = y0; rm.group = h; rm.nparts = 1; rm.part = 0;
    hipError_t e = srx_launch_raygen(cam, x0, w, rm, 0, h, pass_index, seed, 0, nullptr, d_d, d_k, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(dir3, d_d, n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(keys, d_k, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_d); (void)hipFree(d_k);
    if (e != hipSuccess) SRX_FAIL(std::string("srx_camera_rays: ") + hipGetErrorString(e));
    return 0;
}

// Camera::random_slope_at (Camera.h:91-97, :138-140) with the reference's own generator: ONE XorGen (XorRandom.cpp) seeded
// with seed4, two fRand draws per pixel, pixels in row-major order of the rectangle. The parity ray sets of SURVEY 8(d).
extern "C" int srx_camera_rays_xorgen(const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h, uint32_t pass_index,
                                      const uint64_t* seed4, float* dir3) {
    if (!camera || !seed4 || !dir3) SRX_FAIL("srx_camera_rays_xorgen: null argument");
    if (camera->res_x <= 0 || camera->res_y <= 0 || camera->subdiv <= 0) SRX_FAIL("srx_camera_rays_xorgen: bad camera");
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > camera->res_x || y0 + h > camera->res_y)
        SRX_FAIL("srx_camera_rays_xorgen: rectangle outside the camera resolution");
    if ((uint64_t)w * h > (1u << 22)) SRX_FAIL("srx_camera_rays_xorgen: at most 2^22 rays (one sequential generator stream)");
    GCamera cam;
    std::memcpy(&cam, camera, sizeof(cam));
    const size_t n = (size_t)w * h;
    float* d_d = nullptr;
    SRX_HIP(hipMalloc((void**)&d_d, n * 12));
    hipError_t e = srx_launch_raygen_xorgen(cam, x0, y0, w, h, pass_index, seed4, d_d, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(dir3, d_d, n * 12, hipMemcpyDeviceToHost);
    (void)hipFree(d_d);
    if (e != hipSuccess) SRX_FAIL(std::string("srx_camera_rays_xorgen: ") + hipGetErrorString(e));
    return 0;
}

// XorGen streams from the device (XorRandom.cpp:35-69 pools, :119-132 scalars), pools first: for parity tests of the generator itself.
extern "C" int srx_xorgen_stream(const uint64_t* seed4, uint32_t n_pools, uint32_t n_scalar, float* pools8, float* scalars) {
    if (!seed4 || (n_pools && !pools8) || (n_scalar && !scalars)) SRX_FAIL("srx_xorgen_stream: null argument");
    if ((uint64_t)n_pools + n_scalar > (1u << 22)) SRX_FAIL("srx_xorgen_stream: at most 2^22 draws");
    float *d_p = nullptr, *d_s = nullptr;
    hipError_t e = hipMalloc((void**)&d_p, std::max<size_t>(1, (size_t)n_pools * 32));
    if (e == hipSuccess) e = hipMalloc((void**)&d_s, std::max<size_t>(1, (size_t)n_scalar * 4));
    if (e == hipSuccess) e = srx_launch_xorgen_stream(seed4, n_pools, n_scalar, d_p, d_s, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && n_pools) e = hipMemcpy(pools8, d_p, (size_t)n_pools * 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_scalar) e = hipMemcpy(scalars, d_s, (size_t)n_scalar * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_p); (void)hipFree(d_s);
    if (e != hipSuccess) SRX_FAIL(std::string("srx_xorgen_stream: ") + hipGetErrorString(e));
    return 0;
}
