// oracle/sro_noise.h — CPU restatement of the adaptive-sampling controller.
// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). PARITY UNPINNED upstream: the reference holds no fixture for it.
//
// Follows SceneManager::analyze_noise (SceneManager.h:526-667), the priority part of
// SceneManager::balance_threads (:668-692) and the per-block visit rule of
// RenderThread::process_iteratively (:193-198). Expression order, float/double/int promotions and the
// running-min quirk of the first loop are kept. Two deliberate differences, shared with the product:
//  * the tie-break draw `gen.fRand(-1, 1)` (:639) comes from a keyed stream addressed by the region's rectangle
//    and the analysis index (the reference uses the calling thread's clock-seeded generator);
//  * a split at offset 0 is refused: the reference would push an empty region and re-split the same rectangle
//    for ever when every column (or row) score is 0 (:596-599 start at 0 and nothing beats 0).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>
#include "sro_math.h"

namespace sro {

struct NoiseRegion {                       // SceneManager.h:262-273
    int x0, y0, x1, y1;
    bool active = true;
    float noise = 1, max_noise = 1, min_noise = 1, avg_noise = 0;
    NoiseRegion(int ax0, int ay0, int ax1, int ay1) : x0(ax0), y0(ay0), x1(ax1), y1(ay1) {}
};

// ImageHandler::luminance (ImageProcessing.h:115-118) = XYZ::dot(v, (0.2126, 0.7152, 0.0722)) in float
inline float noise_luminance(float r, float g, float b) { return r * 0.2126f + g * 0.7152f + b * 0.0722f; }

// per-pixel term of SceneManager.h:568-572; 0 where the reference `continue`s (main luminance 0)
// abs_luminance: the product's opt-in SRX_NOISE_ABS_LUMINANCE (NOT the reference): |L| in the denominator
inline float noise_pixel_error(const float* main4, const float* aux4, bool abs_luminance = false) {
    const float main_lum = noise_luminance(main4[0], main4[1], main4[2]);
    const float aux_lum = noise_luminance(aux4[0], aux4[1], aux4[2]);
    const float delta = std::fabs(main_lum - aux_lum);
    if (main_lum == 0) return 0.0f;
    const float den_lum = abs_luminance ? std::fabs(main_lum) : main_lum;
    return delta / std::max(std::pow(den_lum, 1.1f), den_lum);
}

// keyed uniform in [lo, hi): one draw addressed by (seed, a, b), conversions of XorRandom.cpp:119-121
inline float noise_keyed_rand(uint64_t seed, uint64_t a, uint64_t b, float lo, float hi) {
    KeyedGen g;
    g.s = KeyedGen::mix(seed ^ KeyedGen::mix(a + 0x2545F4914F6CDD1Dull) ^ KeyedGen::mix(b + 0x9E6C63D0676A9A99ull));
    return g.fRand(lo, hi);
}

struct NoiseController {
    int res_x = 0, res_y = 0;
    uint64_t seed = 0;
    uint32_t analyses = 0;
    int minimum_samples = 64;              // SceneManager.h:301
    std::vector<NoiseRegion> regions;
    static constexpr int block_size = 4;   // SceneManager.h:299

    void reset(int rx, int ry, uint64_t sd) {   // SceneManager.h:937-938
        res_x = rx; res_y = ry; seed = sd; analyses = 0;
        regions.clear();
        regions.emplace_back(0, 0, rx, ry);
    }

    // SceneManager.h:526-667. err = per-pixel error map (res_y x res_x), seen = rStat.minimum_samples (:492-501)
    void analyze(const float* err, int seen) {
        const int minimum_size = 8;
        const double split_threshold = 0.27;
        const double done_threshold = 0.00001;
        float max_noise = 0;
        float min_noise = regions[0].noise;
        float avg_noise = 0;
        int area_total = 0;
        for (size_t i = 0; i < regions.size(); i++) {                 // :535-547
            NoiseRegion& NR = regions[i];
            max_noise = std::max(NR.noise, max_noise);
            if (NR.active && NR.noise > 0) {
                float area = (float)((NR.x1 - NR.x0) * (NR.y1 - NR.y0));
                min_noise = std::min(NR.noise, min_noise);
                avg_noise *= ((float)area_total / ((float)area_total + area));
                avg_noise += std::min(NR.noise, min_noise) * (area / ((float)area_total + area));
                area_total = (int)((float)area_total + area);
            }
        }
        if (avg_noise == 0) avg_noise = 1;
        std::vector<float> col_error, row_error;
        for (int i = 0; i < (int)regions.size(); i++) {               // :552-666
            NoiseRegion& NR = regions[i];
            NR.max_noise = max_noise; NR.min_noise = min_noise; NR.avg_noise = avg_noise;
            const int dX = NR.x1 - NR.x0, dY = NR.y1 - NR.y0;
            const double count = dX * dY;
            double total_error = 0;
            col_error.assign(dX, 0.0f);
            row_error.assign(dY, 0.0f);
            const float col_scalar = (float)(1.0 / dY);
            const float row_scalar = (float)(1.0 / dX);
            for (int x = NR.x0; x < NR.x1; x++) {
                for (int y = NR.y0; y < NR.y1; y++) {
                    const float error = err[(size_t)y * res_x + x];
                    col_error[x - NR.x0] += error * col_scalar;
                    row_error[y - NR.y0] += error * row_scalar;
                    total_error += error / count;
                }
            }
            NR.noise = (float)total_error;
            if (!NR.active) continue;
            if (seen >= minimum_samples) {
                if (total_error < done_threshold) {
                    if (dX * dY <= (minimum_size * minimum_size * 2) || total_error == 0) { NR.active = false; continue; }
                } else {
                    NR.active = true;
                }
                if (total_error < split_threshold) {
                    if (dX / 2 < minimum_size && dY / 2 < minimum_size) continue;
                    int best_col = 0, best_row = 0;
                    float best_col_score = 0, best_row_score = 0;
                    float bin1 = 0; int bin1_size = 0;
                    float bin2 = (float)total_error; int bin2_size = dX;
                    for (int ci = 0; ci < dX - 1; ci++) {             // :605-619
                        bin1 *= ((float)bin1_size / (bin1_size + 1));
                        bin2 *= ((float)bin2_size / (bin2_size - 1));
                        bin1 += col_error[ci] / (bin1_size + 1);
                        bin2 -= col_error[ci] / (bin2_size - 1);
                        bin1_size++; bin2_size--;
                        const float score = (float)(std::fabs(bin1 - bin2) / total_error);
                        if (ci >= minimum_size && ci <= dX - minimum_size && score > best_col_score) { best_col_score = score; best_col = ci; }
                    }
                    bin1 = 0; bin1_size = 0; bin2 = (float)total_error; bin2_size = dY;
                    for (int ri = 0; ri < dY - 1; ri++) {             // :624-638
                        bin1 *= ((float)bin1_size / (bin1_size + 1));
                        bin2 *= ((float)bin2_size / (bin2_size - 1));
                        bin1 += row_error[ri] / (bin1_size + 1);
                        bin2 -= row_error[ri] / (bin2_size - 1);
                        bin1_size++; bin2_size--;
                        const float score = (float)(std::fabs(bin1 - bin2) / total_error);
                        if (ri >= minimum_size && ri <= dY - minimum_size && score > best_row_score) { best_row_score = score; best_row = ri; }
                    }
                    if (std::fabs(best_col_score - best_row_score) < 0.000001)     // :639
                        best_col_score += noise_keyed_rand(seed, ((uint64_t)(uint32_t)NR.x0 << 32) | (uint32_t)NR.y0,
                                                           (((uint64_t)(uint32_t)NR.x1 << 32) | (uint32_t)NR.y1) ^ ((uint64_t)analyses << 48), -1, 1);
                    const bool can_slice_y = (dY / 2 >= minimum_size);
                    const bool can_slice_x = (dX / 2 >= minimum_size);
                    if ((can_slice_x && best_col_score > best_row_score) || !can_slice_y) {      // :642-652
                        if (best_col == 0) continue;                  // refused, see header
                        NoiseRegion nr(NR.x0 + best_col, NR.y0, NR.x1, NR.y1);
                        NR.x1 = NR.x0 + best_col;
                        regions.push_back(nr);                        // invalidates NR
                        i--;
                        continue;
                    }
                    if ((can_slice_y && best_row_score > best_col_score) || !can_slice_x) {      // :653-663
                        if (best_row == 0) continue;
                        NoiseRegion nr(NR.x0, NR.y0 + best_row, NR.x1, NR.y1);
                        NR.y1 = NR.y0 + best_row;
                        regions.push_back(nr);
                        i--;
                        continue;
                    }
                }
            }
        }
        analyses++;
    }

    int blocks_x() const { return (res_x + block_size - 1) / block_size; }
    int blocks_y() const { return (res_y + block_size - 1) / block_size; }

    // SceneManager.h:669-692: per-pixel noise map, then one priority per 4x4 block. out[by * blocks_x + bx].
    // Returns false (and writes nothing) while seen < minimum_samples (:669): priorities stay at their initial 1.
    bool priorities(int seen, float* out) const {
        if (seen < minimum_samples) return false;
        std::vector<float> noise_map((size_t)res_x * res_y, 0.0f);    // :370-377 rows start at 0
        for (const NoiseRegion& NR : regions)
            for (int y = NR.y0; y < NR.y1; y++)
                for (int x = NR.x0; x < NR.x1; x++)
                    noise_map[(size_t)y * res_x + x] = std::min((float)NR.active * (NR.noise / NR.avg_noise), 4.0f);
        const int total_pixels = block_size * block_size;
        const float multiplier = 1.0f / (total_pixels);
        for (int by = 0; by < blocks_y(); by++) {
            for (int bx = 0; bx < blocks_x(); bx++) {
                float avg_priority = 0;
                for (int pi = 0; pi < total_pixels; pi++) {           // get_pixel_offset :774-776
                    const int x = bx * block_size + pi % block_size, y = by * block_size + (block_size - pi / block_size - 1);
                    const float nm = (x < res_x && y < res_y) ? noise_map[(size_t)y * res_x + x] : 0.0f;
                    avg_priority = (float)(avg_priority + std::pow((double)(nm * multiplier), 2));
                }
                out[(size_t)by * blocks_x() + bx] = avg_priority;
            }
        }
        return true;
    }
};

// RenderThread::process_iteratively :194-198 for one visit of block `block`: does the block take a sample?
// (priority <= 1 by construction of `priorities`, so to_do is 0 or 1; larger priorities are treated as 1 sample.)
inline bool noise_block_fires(float priority, uint64_t seed, uint32_t visit_index, uint32_t block, uint32_t n_blocks) {
    const float scaled = priority;
    const float u = noise_keyed_rand(seed, (uint64_t)visit_index * n_blocks + block, 0x51ED270B0BADC0DEull, 0, 1);
    const bool add_one = u < (scaled - std::floor(scaled));
    const int to_do = (int)(add_one + std::floor(scaled));
    return to_do > 0;
}

}  // namespace sro
