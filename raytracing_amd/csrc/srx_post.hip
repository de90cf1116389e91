// raytracing_amd/csrc/srx_post.hip — output stage: the canvas conversion of the reference's
// SceneManager::refresh_canvas (SceneManager.h:1064-1072) = ImageHandler::post_process_pixel per pixel
// (ImageProcessing.h:137-151) + GUI::commit_pixel's float -> 8-bit store (GUI.h:104-107), and the .raw dump of
// FileManager::writeRaw / readRaw (FileLoading.h:166-199). One streaming kernel over the float4 framebuffer (16 B in,
// 4 B + optional 12 B out per pixel: HBM-bound, a 4K frame is 133 MB in); files are written from a host copy.
//
// The OpenColorIO view transform of :141-143 needs a config file outside the reference's repository: SRX_TONEMAP_NONE
// leaves it out (the parity path), SRX_TONEMAP_AGX_APPROX is the public analytic AgX approximation in its place.
#include <cmath>
#include <cstdio>
#include <cstring>
#include "srx_host.h"

namespace {

struct P3 { float x, y, z; };

__device__ __forceinline__ float post_luminance(P3 v) { return v.x * 0.2126f + v.y * 0.7152f + v.z * 0.0722f; }   // :115-118

// pow(lums, 1.0 / gamma) in double as the reference calls it. The kernel is bound by this double arithmetic, not by HBM:
// gamma == 1 needs no call (x^1 = x exactly), and for lums > 0 the power is exp2(y * log2(x)) in double, whose error
// (~1e-14 relative) reaches the float result only on a rounding tie; everything else takes the library's pow.
__device__ __forceinline__ float post_pow(float lums, float gamma) {
    if (gamma == 1.0f) return lums;
    const double y = 1.0 / (double)gamma;
    if (lums > 0.0f && lums < __builtin_inff()) return (float)exp2(y * log2((double)lums));
    return (float)pow((double)lums, y);
}

__device__ __forceinline__ P3 post_apply_gamma(P3 in, float gamma) {                                             // :128-133
    const float lums = post_luminance(in);
    if (lums == 0.0f) return P3{0.0f, 0.0f, 0.0f};
    const float post_lums = post_pow(lums, gamma);
    return P3{in.x * post_lums / lums, in.y * post_lums / lums, in.z * post_lums / lums};
}

__device__ __forceinline__ P3 post_rgb2hsv(P3 in) {                                                              // :12-55
    P3 out;
    double mn = in.x < in.y ? in.x : in.y;
    mn = mn < in.z ? mn : in.z;
    double mx = in.x > in.y ? in.x : in.y;
    mx = mx > in.z ? mx : in.z;
    out.z = (float)mx;
    const double delta = mx - mn;
    if (delta < 0.00001) { out.y = 0.0f; out.x = 0.0f; return out; }
    if (mx > 0.0) out.y = (float)(delta / mx);
    else { out.y = 0.0f; out.x = __builtin_nanf(""); return out; }
    if (in.x >= mx) out.x = (float)((in.y - in.z) / delta);
    else if (in.y >= mx) out.x = (float)(2.0 + (in.z - in.x) / delta);
    else out.x = (float)(4.0 + (in.x - in.y) / delta);
    out.x = (float)(out.x * 60.0);
    if (out.x < 0.0) out.x = (float)(out.x + 360.0);
    return out;
}

__device__ __forceinline__ P3 post_hsv2rgb(P3 in) {                                                              // :58-114
    if (in.y <= 0.0) return P3{in.z, in.z, in.z};
    double hh = in.x;
    if (hh >= 360.0) hh = 0.0;
    hh /= 60.0;
    const long i = (long)hh;
    const double ff = hh - (double)i;
    const double p = in.z * (1.0 - in.y);
    const double q = in.z * (1.0 - (in.y * ff));
    const double t = in.z * (1.0 - (in.y * (1.0 - ff)));
    switch (i) {
    case 0: return P3{in.z, (float)t, (float)p};
    case 1: return P3{(float)q, in.z, (float)p};
    case 2: return P3{(float)p, in.z, (float)t};
    case 3: return P3{(float)p, (float)q, in.z};
    case 4: return P3{(float)t, (float)p, in.z};
    default: return P3{in.z, (float)p, (float)q};
    }
}

__device__ __forceinline__ P3 post_agx_approx(P3 v) {
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    float c[3] = {0.842479062253094f * v.x + 0.0784335999999992f * v.y + 0.0792237451477643f * v.z,
                  0.0423282422610123f * v.x + 0.878468636469772f * v.y + 0.0791661274605434f * v.z,
                  0.0423756549057051f * v.x + 0.0784336f * v.y + 0.879142973793104f * v.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float x = log2f(c[k] > 1e-10f ? c[k] : 1e-10f);
        x = x > min_ev ? x : min_ev;
        x = x < max_ev ? x : max_ev;
        x = (x - min_ev) / (max_ev - min_ev);
        const float x2 = x * x, x4 = x2 * x2;
        c[k] = 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
    }
    return P3{1.19687900512017f * c[0] - 0.0980208811401368f * c[1] - 0.0990297440797205f * c[2],
              -0.0528968517574562f * c[0] + 1.15190312990417f * c[1] - 0.0989611768448433f * c[2],
              -0.0529716355144438f * c[0] - 0.0980434501171241f * c[1] + 1.15107367264116f * c[2]};
}

// std::min(std::max(v, 0), 1) (XYZ.cpp:306-315): a NaN stays a NaN
__device__ __forceinline__ float post_clamp01(float v) {
    const float a = (v < 0.0f) ? 0.0f : v;
    return (1.0f < a) ? 1.0f : a;
}
__device__ __forceinline__ uint32_t post_to_u8(float v) { return (v != v) ? 0u : (uint32_t)(int)v; }   // GUI.h:104-107

__global__ __launch_bounds__(256) void k_post_process(const float4* __restrict__ accum, uint32_t n, int tonemap,
                                                      uint32_t* __restrict__ rgba8, float* __restrict__ rgb_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    P3 c{a.x, a.y, a.z};
    c = post_apply_gamma(c, 1.0f);
    if (tonemap == SRX_TONEMAP_AGX_APPROX) c = post_agx_approx(c);
    c = post_apply_gamma(c, 2.2f);
    c = post_hsv2rgb(post_rgb2hsv(c));
    const float r = post_clamp01(c.x) * 255.0f, g = post_clamp01(c.y) * 255.0f, b = post_clamp01(c.z) * 255.0f;
    if (rgba8) rgba8[i] = post_to_u8(r) | (post_to_u8(g) << 8) | (post_to_u8(b) << 16) | 0xFF000000u;
    if (rgb_out) { rgb_out[3 * (size_t)i] = r; rgb_out[3 * (size_t)i + 1] = g; rgb_out[3 * (size_t)i + 2] = b; }
}

// ---- PNG (8-bit RGBA, no interlace) with stored deflate blocks: no compression library needed ----------------
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
void put_chunk(std::vector<uint8_t>& png, const char* type, const std::vector<uint8_t>& data) {
    put_be32(png, (uint32_t)data.size());
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_update(0xFFFFFFFFu, png.data() + at, png.size() - at) ^ 0xFFFFFFFFu);
}

}  // namespace

extern "C" int srx_post_process(const float* d_accum4, int32_t res_x, int32_t res_y, int32_t tonemap, uint8_t* d_rgba8,
                                float* d_rgb_out) {
    if (!d_accum4 || (!d_rgba8 && !d_rgb_out)) SRX_FAIL("srx_post_process: null argument");
    if (res_x <= 0 || res_y <= 0 || (int64_t)res_x * res_y >= (1ll << 31)) SRX_FAIL("srx_post_process: bad resolution");
    if (tonemap != SRX_TONEMAP_NONE && tonemap != SRX_TONEMAP_AGX_APPROX) SRX_FAIL("srx_post_process: unknown tonemap");
    const uint32_t n = (uint32_t)res_x * (uint32_t)res_y;
    hipLaunchKernelGGL(k_post_process, dim3((n + 255) / 256), dim3(256), 0, 0, (const float4*)d_accum4, n, tonemap, (uint32_t*)d_rgba8,
                       d_rgb_out);
    SRX_HIP(hipGetLastError());
    SRX_HIP(hipDeviceSynchronize());
    return 0;
}

// ---- device buffers for hosts without a HIP toolchain (the reference keeps its framebuffers on the host:
// SceneManager::raw_output / aux_output, SceneManager.h:288-289, 370-380)
extern "C" int srx_buffer_create(uint64_t bytes, void** d_out) {
    if (!d_out || bytes == 0) SRX_FAIL("srx_buffer_create: null argument or zero size");
    void* p = nullptr;
    SRX_HIP(hipMalloc(&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(p); SRX_FAIL(std::string("srx_buffer_create: ") + hipGetErrorString(e)); }
    *d_out = p;
    return 0;
}
extern "C" void srx_buffer_destroy(void* d_buf) { if (d_buf) (void)hipFree(d_buf); }
extern "C" int srx_buffer_read(const void* d_buf, uint64_t bytes, void* host_out) {
    if (!d_buf || !host_out) SRX_FAIL("srx_buffer_read: null argument");
    SRX_HIP(hipMemcpy(host_out, d_buf, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int srx_buffer_write(void* d_buf, uint64_t bytes, const void* host_in) {
    if (!d_buf || !host_in) SRX_FAIL("srx_buffer_write: null argument");
    SRX_HIP(hipMemcpy(d_buf, host_in, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int srx_buffer_clear(void* d_buf, uint64_t bytes) {
    if (!d_buf) SRX_FAIL("srx_buffer_clear: null argument");
    SRX_HIP(hipMemset(d_buf, 0, bytes));
    return 0;
}

extern "C" int srx_write_raw(const char* path, const float* d_accum4, int32_t res_x, int32_t res_y) {
    if (!path || !d_accum4) SRX_FAIL("srx_write_raw: null argument");
    if (res_x <= 0 || res_y <= 0) SRX_FAIL("srx_write_raw: bad resolution");
    const size_t n = (size_t)res_x * (size_t)res_y;
    std::vector<float> host(n * 4);
    SRX_HIP(hipMemcpy(host.data(), d_accum4, n * 16, hipMemcpyDeviceToHost));
    std::vector<float> rgb(n * 3);
    for (size_t i = 0; i < n; i++) { rgb[3 * i] = host[4 * i]; rgb[3 * i + 1] = host[4 * i + 1]; rgb[3 * i + 2] = host[4 * i + 2]; }
    FILE* f = std::fopen(path, "wb");
    if (!f) SRX_FAIL(std::string("srx_write_raw: cannot open ") + path);
    const uint32_t hdr[2] = {(uint32_t)res_x, (uint32_t)res_y};                      // FileLoading.h:172-176
    bool ok = std::fwrite(hdr, 4, 2, f) == 2 && std::fwrite(rgb.data(), 12, n, f) == n;   // :177-181, row 0 first
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) SRX_FAIL(std::string("srx_write_raw: short write to ") + path);
    return 0;
}

extern "C" int srx_read_raw(const char* path, float* rgb, uint64_t capacity_pixels, int32_t* res_x, int32_t* res_y) {
    if (!path || !res_x || !res_y) SRX_FAIL("srx_read_raw: null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) SRX_FAIL(std::string("srx_read_raw: cannot open ") + path);
    int32_t hdr[2] = {0, 0};                                                          // FileLoading.h:185-188
    if (std::fread(hdr, 4, 2, f) != 2 || hdr[0] <= 0 || hdr[1] <= 0) { std::fclose(f); SRX_FAIL("srx_read_raw: bad header"); }
    *res_x = hdr[0]; *res_y = hdr[1];
    const uint64_t n = (uint64_t)hdr[0] * (uint64_t)hdr[1];
    if (rgb) {
        if (n > capacity_pixels) { std::fclose(f); SRX_FAIL("srx_read_raw: buffer too small"); }
        if (std::fread(rgb, 12, n, f) != n) { std::fclose(f); SRX_FAIL("srx_read_raw: file shorter than its header says"); }
    }
    std::fclose(f);
    return 0;
}

extern "C" int srx_write_png(const char* path, const uint8_t* d_rgba8, int32_t res_x, int32_t res_y) {
    if (!path || !d_rgba8) SRX_FAIL("srx_write_png: null argument");
    if (res_x <= 0 || res_y <= 0) SRX_FAIL("srx_write_png: bad resolution");
    const size_t row = (size_t)res_x * 4, n = row * (size_t)res_y;
    std::vector<uint8_t> img(n);
    SRX_HIP(hipMemcpy(img.data(), d_rgba8, n, hipMemcpyDeviceToHost));
    std::vector<uint8_t> raw;                       // filter byte 0 + row
    raw.reserve(n + (size_t)res_y);
    for (int32_t y = 0; y < res_y; y++) { raw.push_back(0); raw.insert(raw.end(), img.begin() + y * row, img.begin() + (y + 1) * row); }
    std::vector<uint8_t> z;                         // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    for (size_t at = 0; at < raw.size(); at += 65535) {
        const size_t len = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + len == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(len & 0xFF)); z.push_back((uint8_t)(len >> 8));
        z.push_back((uint8_t)(~len & 0xFF)); z.push_back((uint8_t)((~len >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + len);
        for (size_t i = at; i < at + len; i++) { s1 = (s1 + raw[i]) % 65521u; s2 = (s2 + s1) % 65521u; }
    }
    put_be32(z, (s2 << 16) | s1);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)res_x); put_be32(ihdr, (uint32_t)res_y);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) SRX_FAIL(std::string("srx_write_png: cannot open ") + path);
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) SRX_FAIL(std::string("srx_write_png: short write to ") + path);
    return 0;
}
