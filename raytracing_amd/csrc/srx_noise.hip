// raytracing_amd/csrc/srx_noise.hip — C-ABI: adaptive-sampling controller.
//
// Replaces SceneManager::analyze_noise (SceneManager.h:526-667) and the priority half of
// SceneManager::balance_threads (:668-692). Split of the work:
//   device  k_noise_error    per-pixel error |L - L_aux| / max(L^1.1, L) (:568-572) + minimum sample count (:492-501)
//           k_region_stats   per region: column sums, row sums (float, the reference's accumulation order) and the
//                            total (double, per column then across columns), one workgroup per region, a batch per launch
//           k_paint_regions, k_block_priority   noise map (:670-676) and the 4x4-block priorities (:679-692)
//   host    the region list and the split search (:582-665): a few hundred scalar steps per region whose ORDER is
//           load-bearing (the list order feeds the running-min average of the next analysis, :535-547), driven by
//           the measured sums; every region that exists when the analysis starts is measured in one launch, the two
//           halves of a split are summed on the host in the same order (no launch per split).
// Differences from the reference (shared with the CPU oracle): the tie-break draw of :639 is keyed by the region's
// rectangle and the analysis index; a split at offset 0 is refused (the reference would loop for ever on it).
#include "srx_host.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

struct Rect { int x0, y0, x1, y1; };
struct RectJob { int x0, y0, x1, y1; unsigned int col_off, row_off; };   // offsets into the float / double outputs

__device__ __forceinline__ float lum3(float r, float g, float b) { return r * 0.2126f + g * 0.7152f + b * 0.0722f; }   // ImageProcessing.h:115-118

// owned rows (GRows) get their error, the other rows 0, so an all-reduce(sum) over the GPUs completes the map
__global__ __launch_bounds__(256) void k_noise_error(const float4* accum4, const float4* aux4, int res_x, int res_y, GRows rm, float* err,
                                                    int* min_samples, int abs_luminance) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)res_x * (uint32_t)res_y;
    int mine = 0x7fffffff;
    if (idx < n) {
        const int y = (int)(idx / (uint32_t)res_x);
        const bool owned = srx_row_owned(rm, y);
        float e = 0.0f;
        if (owned) {
            const float4 m = accum4[idx], a = aux4[idx];
            const float main_lum = lum3(m.x, m.y, m.z), aux_lum = lum3(a.x, a.y, a.z);
            const float delta = fabsf(main_lum - aux_lum);
            // :571-572. A negative main_lum makes powf NaN, the region's noise NaN and every priority NaN (quirk Q11);
            // abs_luminance (opt-in, NOT the reference) takes |main_lum| in the denominator instead
            const float den_lum = abs_luminance ? fabsf(main_lum) : main_lum;
            if (main_lum != 0.0f) e = delta / std_max(powf(den_lum, 1.1f), den_lum);
            mine = (int)m.w;
        }
        err[idx] = e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine = min(mine, __shfl_xor(mine, off, 64));
    if ((threadIdx.x & 63) == 0 && mine != 0x7fffffff) atomicMin(min_samples, mine);
}

// SceneManager.h:557-577 for a batch of regions: one workgroup per region
__global__ __launch_bounds__(256) void k_region_stats(const float* err, int res_x, const RectJob* jobs, float* cols, float* rows,
                                                     double* col_totals, double* totals) {
    const RectJob j = jobs[blockIdx.x];
    const int dX = j.x1 - j.x0, dY = j.y1 - j.y0;
    const double count = (double)(dX * dY);
    const float col_scalar = (float)(1.0 / dY);
    const float row_scalar = (float)(1.0 / dX);
    for (int x = j.x0 + (int)threadIdx.x; x < j.x1; x += (int)blockDim.x) {
        float acc = 0.0f;
        double tot = 0.0;
        for (int y = j.y0; y < j.y1; y++) {
            const float e = err[(size_t)y * res_x + x];
            acc += e * col_scalar;
            tot += (double)e / count;
        }
        cols[j.col_off + (x - j.x0)] = acc;
        col_totals[j.col_off + (x - j.x0)] = tot;
    }
    for (int y = j.y0 + (int)threadIdx.x; y < j.y1; y += (int)blockDim.x) {
        float acc = 0.0f;
        for (int x = j.x0; x < j.x1; x++) acc += err[(size_t)y * res_x + x] * row_scalar;
        rows[j.row_off + (y - j.y0)] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int c = 0; c < dX; c++) total += col_totals[j.col_off + c];
        totals[blockIdx.x] = total;
    }
}

struct PaintJob { int x0, y0, x1, y1; float value; };
__global__ __launch_bounds__(256) void k_paint_regions(const PaintJob* jobs, int res_x, float* noise_map) {   // :670-676
    const PaintJob j = jobs[blockIdx.x];
    const int dX = j.x1 - j.x0, n = dX * (j.y1 - j.y0);
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) noise_map[(size_t)(j.y0 + i / dX) * res_x + (j.x0 + i % dX)] = j.value;
}
__global__ __launch_bounds__(256) void k_block_priority(const float* noise_map, int res_x, int res_y, int blocks_x, int blocks_y, float* priority) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= blocks_x * blocks_y) return;
    const int bx = b % blocks_x, by = b / blocks_x;
    const float multiplier = 1.0f / 16;
    float avg_priority = 0.0f;
    for (int pi = 0; pi < 16; pi++) {                                  // get_pixel_offset :774-776
        const int x = bx * 4 + pi % 4, y = by * 4 + (4 - pi / 4 - 1);
        const float nm = (x < res_x && y < res_y) ? noise_map[(size_t)y * res_x + x] : 0.0f;
        const double v = (double)(nm * multiplier);
        avg_priority = (float)((double)avg_priority + v * v);          // pow(x, 2) :688
    }
    priority[b] = avg_priority;
}
__global__ __launch_bounds__(256) void k_fill(float* p, uint32_t n, float v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Stats { std::vector<float> col, row; double total = 0; };

uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// one keyed uniform draw in [lo, hi): same construction as the device's KeyedGen::fRand
float keyed_rand(uint64_t seed, uint64_t a, uint64_t b, float lo, float hi) {
    uint64_t s = mix64(seed ^ mix64(a + 0x2545F4914F6CDD1Dull) ^ mix64(b + 0x9E6C63D0676A9A99ull));
    s += 0x9E3779B97F4A7C15ull;
    const float f = (float)((double)(mix64(s) >> 11) * 0x1.0p-53);
    return lo + f * (hi - lo);
}

}  // namespace

struct srx_noise {
    int res_x = 0, res_y = 0;
    uint64_t seed = 0;
    uint32_t analyses = 0;
    int minimum_samples = 64;                     // SceneManager.h:301
    int abs_luminance = 0;                        // SRX_NOISE_ABS_LUMINANCE (opt-in, not the reference)
    std::vector<srx_noise_region> regions;
    // device scratch, grown on demand
    RectJob* d_jobs = nullptr; size_t jobs_cap = 0;
    float* d_cols = nullptr; float* d_rows = nullptr; double* d_col_totals = nullptr; size_t cols_cap = 0, rows_cap = 0, col_totals_cap = 0;
    double* d_totals = nullptr; size_t totals_cap = 0;
    float* d_noise_map = nullptr;
    int* d_min = nullptr;
};

namespace {

int grow(void** p, size_t* cap, size_t need, size_t elem) {
    if (*cap >= need) return 0;
    if (*p) SRX_HIP(hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t n = std::max<size_t>(need * 2, 1024);
    SRX_HIP(hipMalloc(p, n * elem));
    *cap = n;
    return 0;
}

int measure(srx_noise* c, const float* d_err, const std::vector<Rect>& rects, std::vector<Stats>& out) {
    out.assign(rects.size(), Stats());
    if (rects.empty()) return 0;
    std::vector<RectJob> jobs(rects.size());
    size_t nc = 0, nr = 0;
    for (size_t i = 0; i < rects.size(); i++) {
        const Rect& r = rects[i];
        jobs[i] = RectJob{r.x0, r.y0, r.x1, r.y1, (unsigned)nc, (unsigned)nr};
        nc += (size_t)std::max(0, r.x1 - r.x0); nr += (size_t)std::max(0, r.y1 - r.y0);
    }
    if (grow((void**)&c->d_jobs, &c->jobs_cap, jobs.size(), sizeof(RectJob))) return 1;
    if (grow((void**)&c->d_col_totals, &c->col_totals_cap, nc, sizeof(double))) return 1;
    if (grow((void**)&c->d_cols, &c->cols_cap, nc, sizeof(float))) return 1;
    if (grow((void**)&c->d_rows, &c->rows_cap, nr, sizeof(float))) return 1;
    if (grow((void**)&c->d_totals, &c->totals_cap, jobs.size(), sizeof(double))) return 1;
    SRX_HIP(hipMemcpy(c->d_jobs, jobs.data(), jobs.size() * sizeof(RectJob), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_region_stats, dim3((unsigned)jobs.size()), dim3(256), 0, 0, d_err, c->res_x, c->d_jobs, c->d_cols, c->d_rows,
                       c->d_col_totals, c->d_totals);
    SRX_HIP(hipGetLastError());
    std::vector<float> cols(nc), rows(nr);
    std::vector<double> totals(jobs.size());
    if (nc) SRX_HIP(hipMemcpy(cols.data(), c->d_cols, nc * sizeof(float), hipMemcpyDeviceToHost));
    if (nr) SRX_HIP(hipMemcpy(rows.data(), c->d_rows, nr * sizeof(float), hipMemcpyDeviceToHost));
    SRX_HIP(hipMemcpy(totals.data(), c->d_totals, totals.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < rects.size(); i++) {
        const int dX = std::max(0, rects[i].x1 - rects[i].x0), dY = std::max(0, rects[i].y1 - rects[i].y0);
        out[i].col.assign(cols.begin() + jobs[i].col_off, cols.begin() + jobs[i].col_off + dX);
        out[i].row.assign(rows.begin() + jobs[i].row_off, rows.begin() + jobs[i].row_off + dY);
        out[i].total = (dX && dY) ? totals[i] : 0.0;
    }
    return 0;
}

// best split offset along one axis (SceneManager.h:601-619 / :620-638)
void scan_axis(const std::vector<float>& sums, double total_error, int minimum_size, int* best, float* best_score) {
    const int n = (int)sums.size();
    float bin1 = 0; int bin1_size = 0;
    float bin2 = (float)total_error; int bin2_size = n;
    *best = 0; *best_score = 0;
    for (int i = 0; i < n - 1; i++) {
        bin1 *= ((float)bin1_size / (bin1_size + 1));
        bin2 *= ((float)bin2_size / (bin2_size - 1));
        bin1 += sums[i] / (bin1_size + 1);
        bin2 -= sums[i] / (bin2_size - 1);
        bin1_size++; bin2_size--;
        const float score = (float)(std::fabs(bin1 - bin2) / total_error);
        if (i >= minimum_size && i <= n - minimum_size && score > *best_score) { *best_score = score; *best = i; }
    }
}

}  // namespace

extern "C" int srx_noise_create(int32_t res_x, int32_t res_y, uint64_t seed, int32_t minimum_samples, srx_noise** out) {
    if (!out) SRX_FAIL("srx_noise_create: null argument");
    if (res_x <= 0 || res_y <= 0 || (uint64_t)res_x * res_y > 0x00FFFFFFull) SRX_FAIL("srx_noise_create: bad resolution");
    srx_noise* c = new srx_noise();
    c->res_x = res_x; c->res_y = res_y; c->seed = seed;
    if (minimum_samples >= 0) c->minimum_samples = minimum_samples;
    srx_noise_region whole;
    std::memset(&whole, 0, sizeof(whole));
    whole.x1 = res_x; whole.y1 = res_y; whole.active = 1;             // SceneManager.h:937-938, :265-269
    whole.noise = 1; whole.max_noise = 1; whole.min_noise = 1; whole.avg_noise = 0;
    c->regions.push_back(whole);
    hipError_t e = hipMalloc((void**)&c->d_noise_map, (size_t)res_x * res_y * sizeof(float));
    if (e == hipSuccess) e = hipMemset(c->d_noise_map, 0, (size_t)res_x * res_y * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_min, sizeof(int));
    if (e != hipSuccess) { srx_noise_destroy(c); SRX_FAIL(std::string("srx_noise_create: ") + hipGetErrorString(e)); }
    *out = c;
    return 0;
}

extern "C" int srx_noise_set_option(srx_noise* c, int32_t option, int32_t value) {
    if (!c) SRX_FAIL("srx_noise_set_option: null controller");
    if (option == SRX_NOISE_ABS_LUMINANCE) { c->abs_luminance = value != 0; return 0; }
    SRX_FAIL("srx_noise_set_option: unknown option");
}

extern "C" void srx_noise_destroy(srx_noise* c) {
    if (!c) return;
    (void)hipFree(c->d_jobs); (void)hipFree(c->d_cols); (void)hipFree(c->d_rows); (void)hipFree(c->d_col_totals);
    (void)hipFree(c->d_totals); (void)hipFree(c->d_noise_map); (void)hipFree(c->d_min);
    delete c;
}

namespace {
int noise_error_map(srx_noise* c, const float* d_accum4, const float* d_aux4, const GRows& rm, float* d_err, int32_t* min_samples) {
    const int big = 0x7fffffff;
    SRX_HIP(hipMemcpy(c->d_min, &big, sizeof(int), hipMemcpyHostToDevice));
    const uint32_t n = (uint32_t)c->res_x * (uint32_t)c->res_y;
    hipLaunchKernelGGL(k_noise_error, dim3((n + 255) / 256), dim3(256), 0, 0, (const float4*)d_accum4, (const float4*)d_aux4, c->res_x,
                       c->res_y, rm, d_err, c->d_min, c->abs_luminance);
    SRX_HIP(hipGetLastError());
    int mn = 0;
    SRX_HIP(hipMemcpy(&mn, c->d_min, sizeof(int), hipMemcpyDeviceToHost));
    if (min_samples) *min_samples = (mn == big) ? 0 : mn;
    return 0;
}
}  // namespace

extern "C" int srx_noise_error_map(srx_noise* c, const float* d_accum4, const float* d_aux4, int32_t group_rows, int32_t part,
                                   int32_t nparts, float* d_err, int32_t* min_samples) {
    if (!c || !d_accum4 || !d_aux4 || !d_err) SRX_FAIL("srx_noise_error_map: null argument");
    if (group_rows <= 0 || nparts <= 0 || part < 0 || part >= nparts) SRX_FAIL("srx_noise_error_map: bad row interleave (group_rows, part, nparts)");
    GRows rm;
    rm.y0 = 0; rm.group = group_rows; rm.nparts = nparts; rm.part = part;
    return noise_error_map(c, d_accum4, d_aux4, rm, d_err, min_samples);
}

// the same for the rows `part` owns under a deal (csrc/srx_deal.hip)
extern "C" int srx_noise_error_map_dealt(srx_noise* c, const float* d_accum4, const float* d_aux4, srx_deal* deal, int32_t part, float* d_err,
                                         int32_t* min_samples) {
    if (!c || !d_accum4 || !d_aux4 || !d_err || !deal) SRX_FAIL("srx_noise_error_map_dealt: null argument");
    if (deal->res_x != c->res_x || deal->res_y != c->res_y) SRX_FAIL("srx_noise_error_map_dealt: the deal was made for another frame size");
    GRows rm;
    rm.y0 = 0; rm.group = deal->group; rm.nparts = deal->nparts; rm.part = part;
    if (srx_deal_device(deal, part, &rm.list, &rm.owner, nullptr)) return 1;
    return noise_error_map(c, d_accum4, d_aux4, rm, d_err, min_samples);
}

extern "C" int srx_noise_analyze(srx_noise* c, const float* d_err, int32_t seen) {
    if (!c || !d_err) SRX_FAIL("srx_noise_analyze: null argument");
    const int minimum_size = 8;
    const double split_threshold = 0.27;
    const double done_threshold = 0.00001;
    std::vector<srx_noise_region>& R = c->regions;
    float max_noise = 0;
    float min_noise = R[0].noise;
    float avg_noise = 0;
    int area_total = 0;
    for (size_t i = 0; i < R.size(); i++) {                              // :535-547 (int / float mix as written there)
        const srx_noise_region& NR = R[i];
        max_noise = std::max(NR.noise, max_noise);
        if (NR.active && NR.noise > 0) {
            const float area = (float)((NR.x1 - NR.x0) * (NR.y1 - NR.y0));
            min_noise = std::min(NR.noise, min_noise);
            avg_noise *= ((float)area_total / ((float)area_total + area));
            avg_noise += std::min(NR.noise, min_noise) * (area / ((float)area_total + area));
            area_total = (int)((float)area_total + area);
        }
    }
    if (avg_noise == 0) avg_noise = 1;

    // measured sums, by region index; everything that exists now in one launch
    std::vector<Stats> stats;
    {
        std::vector<Rect> rects(R.size());
        for (size_t i = 0; i < R.size(); i++) rects[i] = Rect{R[i].x0, R[i].y0, R[i].x1, R[i].y1};
        if (measure(c, d_err, rects, stats)) return 1;
    }
    // The two halves of a split are summed on the host from a copy of the map (fetched at the first split): same
    // accumulation order as k_region_stats, bit for bit, without a launch and three copies per split (the first
    // analysis after minimum_samples makes thousands of splits).
    std::vector<float> h_err;
    auto host_stats = [&](const srx_noise_region& NR, Stats& st) {
        const int dX = NR.x1 - NR.x0, dY = NR.y1 - NR.y0;
        st.col.assign((size_t)std::max(dX, 0), 0.0f); st.row.assign((size_t)std::max(dY, 0), 0.0f); st.total = 0;
        if (dX <= 0 || dY <= 0) return;
        const double count = (double)(dX * dY);
        const float col_scalar = (float)(1.0 / dY), row_scalar = (float)(1.0 / dX);
        double total = 0.0;
        for (int x = NR.x0; x < NR.x1; x++) {
            float acc = 0.0f;
            double tot = 0.0;
            for (int y = NR.y0; y < NR.y1; y++) {
                const float e = h_err[(size_t)y * c->res_x + x];
                acc += e * col_scalar;
                tot += (double)e / count;
            }
            st.col[x - NR.x0] = acc;
            total += tot;
        }
        for (int y = NR.y0; y < NR.y1; y++) {
            float acc = 0.0f;
            const float* row = &h_err[(size_t)y * c->res_x];
            for (int x = NR.x0; x < NR.x1; x++) acc += row[x] * row_scalar;
            st.row[y - NR.y0] = acc;
        }
        st.total = total;
    };
    auto split_measure = [&](size_t left, size_t right) -> int {
        if (h_err.empty()) {
            h_err.resize((size_t)c->res_x * c->res_y);
            SRX_HIP(hipMemcpy(h_err.data(), d_err, h_err.size() * sizeof(float), hipMemcpyDeviceToHost));
        }
        host_stats(R[left], stats[left]);
        stats.emplace_back();
        host_stats(R[right], stats.back());
        return 0;
    };
    for (int i = 0; i < (int)R.size(); i++) {                            // :552-666
        R[i].max_noise = max_noise; R[i].min_noise = min_noise; R[i].avg_noise = avg_noise;
        const int dX = R[i].x1 - R[i].x0, dY = R[i].y1 - R[i].y0;
        const double total_error = stats[i].total;
        R[i].noise = (float)total_error;
        if (!R[i].active) continue;
        if (seen < c->minimum_samples) continue;
        if (total_error < done_threshold) {
            if (dX * dY <= (minimum_size * minimum_size * 2) || total_error == 0) { R[i].active = 0; continue; }
        } else {
            R[i].active = 1;
        }
        if (!(total_error < split_threshold)) continue;
        if (dX / 2 < minimum_size && dY / 2 < minimum_size) continue;
        int best_col, best_row;
        float best_col_score, best_row_score;
        scan_axis(stats[i].col, total_error, minimum_size, &best_col, &best_col_score);
        scan_axis(stats[i].row, total_error, minimum_size, &best_row, &best_row_score);
        if (std::fabs(best_col_score - best_row_score) < 0.000001)       // :639, keyed instead of the thread's generator
            best_col_score += keyed_rand(c->seed, ((uint64_t)(uint32_t)R[i].x0 << 32) | (uint32_t)R[i].y0,
                                         (((uint64_t)(uint32_t)R[i].x1 << 32) | (uint32_t)R[i].y1) ^ ((uint64_t)c->analyses << 48), -1, 1);
        const bool can_slice_y = (dY / 2 >= minimum_size);
        const bool can_slice_x = (dX / 2 >= minimum_size);
        srx_noise_region fresh;
        std::memset(&fresh, 0, sizeof(fresh));
        fresh.active = 1; fresh.noise = 1; fresh.max_noise = 1; fresh.min_noise = 1; fresh.avg_noise = 0;
        if ((can_slice_x && best_col_score > best_row_score) || !can_slice_y) {     // :642-652
            if (best_col == 0) continue;                                  // refused: the reference would never terminate
            fresh.x0 = R[i].x0 + best_col; fresh.y0 = R[i].y0; fresh.x1 = R[i].x1; fresh.y1 = R[i].y1;
            R[i].x1 = R[i].x0 + best_col;
        } else if ((can_slice_y && best_row_score > best_col_score) || !can_slice_x) {   // :653-663
            if (best_row == 0) continue;
            fresh.x0 = R[i].x0; fresh.y0 = R[i].y0 + best_row; fresh.x1 = R[i].x1; fresh.y1 = R[i].y1;
            R[i].y1 = R[i].y0 + best_row;
        } else {
            continue;
        }
        R.push_back(fresh);
        if (split_measure((size_t)i, R.size() - 1)) return 1;
        i--;                                                              // the shrunk region is analysed again (:650, :661)
    }
    c->analyses++;
    return 0;
}

extern "C" int srx_noise_regions(const srx_noise* c, srx_noise_region* out, uint32_t capacity, uint32_t* count) {
    if (!c || !count) SRX_FAIL("srx_noise_regions: null argument");
    *count = (uint32_t)c->regions.size();
    if (out) std::memcpy(out, c->regions.data(), std::min<size_t>(capacity, c->regions.size()) * sizeof(srx_noise_region));
    return 0;
}

extern "C" int srx_noise_priorities(srx_noise* c, int32_t seen, float* d_block_priority, int32_t* updated) {
    if (!c || !d_block_priority) SRX_FAIL("srx_noise_priorities: null argument");
    const int bx = (c->res_x + 3) / 4, by = (c->res_y + 3) / 4;
    const uint32_t nb = (uint32_t)bx * (uint32_t)by;
    if (seen < c->minimum_samples) {                                     // :669: priorities keep their initial 1 (:36)
        hipLaunchKernelGGL(k_fill, dim3((nb + 255) / 256), dim3(256), 0, 0, d_block_priority, nb, 1.0f);
        SRX_HIP(hipGetLastError());
        SRX_HIP(hipDeviceSynchronize());
        if (updated) *updated = 0;
        return 0;
    }
    std::vector<PaintJob> jobs;
    jobs.reserve(c->regions.size());
    for (const srx_noise_region& NR : c->regions) {
        if (NR.x1 <= NR.x0 || NR.y1 <= NR.y0) continue;
        const float v = std::min((float)(NR.active != 0) * (NR.noise / NR.avg_noise), 4.0f);   // :674
        jobs.push_back(PaintJob{NR.x0, NR.y0, NR.x1, NR.y1, v});
    }
    size_t cap_bytes = c->jobs_cap * sizeof(RectJob);
    if (cap_bytes < jobs.size() * sizeof(PaintJob)) {
        if (grow((void**)&c->d_jobs, &c->jobs_cap, (jobs.size() * sizeof(PaintJob) + sizeof(RectJob) - 1) / sizeof(RectJob), sizeof(RectJob))) return 1;
    }
    if (!jobs.empty()) {
        SRX_HIP(hipMemcpy(c->d_jobs, jobs.data(), jobs.size() * sizeof(PaintJob), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_paint_regions, dim3((unsigned)jobs.size()), dim3(256), 0, 0, (const PaintJob*)c->d_jobs, c->res_x, c->d_noise_map);
        SRX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_block_priority, dim3((nb + 255) / 256), dim3(256), 0, 0, c->d_noise_map, c->res_x, c->res_y, bx, by, d_block_priority);
    SRX_HIP(hipGetLastError());
    SRX_HIP(hipDeviceSynchronize());
    if (updated) *updated = 1;
    return 0;
}
