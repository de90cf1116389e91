// raytracing_amd/csrc/srx_scene.hip — C-ABI: device selection, scene upload, batched casts.
//
// srx_scene_create replaces RayEngine::load_scene (Core.h:46-48): the reference keeps a pointer to a
// host PackagedScene (Scene.h:18-31); here the three flat arrays are deep-copied and re-laid for
// 128-byte coalesced fetches (layouts in srx_device.h), textures/materials/lights/tables uploaded,
// and the worst-case traversal stack depth of the tree computed so kernels can size their LDS stack.
#include "srx_host.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>

// ---- scratch of srx_cast_closest_device_static, the entry point that allocates nothing (include/srx.h): the traversal
// stack's overflow areas (SRX_SPILL_SLOTS areas of 8 octets x SRX_STATIC_MAX_STACK entries), their control words and
// the replicated traversal counters, in device memory that comes with the library's code object. Each GPU the library
// runs on holds its own copy (64.1 MB) until the process ends. Host code uses the per-device address taken once with
// hipGetSymbolAddress (static_scratch_for), never the array's name.
#define SRX_STATIC_MAX_STACK 64
#define SRX_STATIC_COUNTER_SLOTS 256
struct StaticScratch {
    unsigned long long area[(size_t)SRX_SPILL_SLOTS * 8 * SRX_STATIC_MAX_STACK];
    uint32_t ctl[16 + SRX_SPILL_SLOTS];
    unsigned long long counters[SRX_STATIC_COUNTER_SLOTS * 2];
};
static __device__ __align__(256) StaticScratch scratch;

static thread_local std::string g_error;
void srx_set_error(const std::string& s) { g_error = s; }

extern "C" const char* srx_last_error(void) { return g_error.c_str(); }

namespace {
// Address of `scratch` on the CURRENT device: taken once per device (at srx_init, or at the first static call on a
// device srx_init was not called for) and kept; every GPU has its own copy of the array at its own address.
enum { SRX_MAX_DEVICES = 64 };
StaticScratch* g_static_scratch[SRX_MAX_DEVICES];
std::mutex g_static_mutex;

int static_scratch_for(int device, StaticScratch** out) {
    if (device < 0 || device >= SRX_MAX_DEVICES) SRX_FAIL("srx: device index beyond the 64 this library keeps static scratch for");
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_scratch[device]) {
        void* p = nullptr;
        SRX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) SRX_FAIL("srx: hipGetSymbolAddress returned no address for the static scratch");
        g_static_scratch[device] = (StaticScratch*)p;
    }
    *out = g_static_scratch[device];
    return 0;
}
}  // namespace

extern "C" int srx_init(int device) {
    int n = 0;
    SRX_HIP(hipGetDeviceCount(&n));
    if (n == 0) SRX_FAIL("srx_init: no HIP device visible; this library has no CPU path");
    if (device < 0 || device >= n) SRX_FAIL("srx_init: device index out of range");
    SRX_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    SRX_HIP(hipGetDeviceProperties(&p, device));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
        SRX_FAIL(std::string("srx_init: kernels are built for gfx950 only, device is ") + p.gcnArchName);
    StaticScratch* sp = nullptr;
    if (static_scratch_for(device, &sp)) return 1;
    // once per device: the traversal kernel's hand-written cross-lane code against its definition (csrc/srx_trace.hip)
    static std::mutex mu;
    static std::vector<int> tested;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(tested.begin(), tested.end(), device) == tested.end()) {
        if (srx_trace_selftest()) return 1;
        tested.push_back(device);
    }
    return 0;
}

namespace {

template <class T>
int upload(srx_scene* s, const T* host, size_t count, const T** out) {
    void* d = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    SRX_HIP(hipMalloc(&d, bytes));
    s->allocations.push_back(d);
    if (count) SRX_HIP(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return 0;
}

inline float rdf(const uint8_t* p, size_t off) { float f; std::memcpy(&f, p + off, 4); return f; }
inline uint32_t rdu(const uint8_t* p, size_t off) { uint32_t u; std::memcpy(&u, p + off, 4); return u; }
inline float u2f_host(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Worst-case number of live stack entries while traversing the tree: visiting a node with m used child
// slots pops 1 entry and pushes at most m; the deepest subtree may be entered first, with the other
// m-1 siblings still below it.
int stack_bound(const uint8_t* nodes, uint32_t n_nodes) {
    std::vector<int> D(n_nodes, 1);
    for (int64_t i = (int64_t)n_nodes - 1; i >= 0; i--) {
        const uint8_t* nd = nodes + (size_t)i * SRX_REF_NODE_BYTES;
        int m = 0, deepest = 0;
        for (int k = 0; k < 8; k++) {
            const uint32_t leaf = rdu(nd, 192 + 4 * k), idx = rdu(nd, 224 + 4 * k);
            if (leaf == 0 && idx == 0) continue;          // unused slot (BVH.h:890-895)
            m++;
            int d = 1;
            if (leaf == 0) d = (idx > (uint32_t)i && idx < n_nodes) ? D[idx] : 1;
            deepest = std::max(deepest, d);
        }
        D[i] = std::max(1, (m - 1) + deepest);
    }
    return n_nodes ? D[0] : 1;
}

}  // namespace

extern "C" int srx_scene_create(const srx_scene_desc* desc, srx_scene** out) {
    if (!desc || !out) SRX_FAIL("srx_scene_create: null argument");
    if (desc->n_nodes == 0 || !desc->nodes) SRX_FAIL("srx_scene_create: scene has no BVH nodes");
    if (2ull * desc->n_nodes + 3ull * desc->n_packets > (unsigned long long)SRX_LINK_LINE_MASK)
        SRX_FAIL("srx_scene_create: nodes + packets exceed 4 GB (2^25 lines of 128 B)");
    if (desc->max_generations < 0 || desc->max_generations >= 32) SRX_FAIL("srx_scene_create: max_generations must be in [0, 31]");
    srx_scene* s = new srx_scene();
    SRX_HIP(hipGetDevice(&s->device));
    std::memset(&s->g, 0, sizeof(GScene));
    GScene& g = s->g;
    const uint8_t* rn = (const uint8_t*)desc->nodes;
    const uint8_t* rp = (const uint8_t*)desc->packets;
    const uint8_t* rt = (const uint8_t*)desc->tris;

    // ---- leaf filter: bounds of the triangles of packets [idx, idx + leaf), and the pad that covers the
    // rounding of the triangle test against that of the slab test: 1e-4 of the scene's size and coordinate range.
    auto packet_bounds = [&](uint32_t first, uint32_t count, float* mn, float* mx) {
        for (int ax = 0; ax < 3; ax++) { mn[ax] = 3.0e38f; mx[ax] = -3.0e38f; }
        for (uint32_t p = first; p < first + count; p++) {
            const uint8_t* pk = rp + (size_t)p * SRX_REF_PACKET_BYTES;
            const int size = (int)(int8_t)pk[420];
            for (int k = 0; k < 8 && k < size; k++) {
                for (int ax = 0; ax < 3; ax++) {
                    const float p1 = rdf(pk, 32 * ax + 4 * k), e1 = rdf(pk, 96 + 32 * ax + 4 * k), e2 = rdf(pk, 192 + 32 * ax + 4 * k);
                    const float v[3] = {p1, p1 + e1, p1 + e2};
                    for (float c : v) { mn[ax] = std::min(mn[ax], c); mx[ax] = std::max(mx[ax], c); }
                }
            }
        }
    };
    // ---- leaf guard: the normal-cone word of the leaf made of packets [first, first + count) (srx_device.h, GNode).
    // The reference's fp32 Moller-Trumbore (Primitives.h:159-202) divides by a det that is mostly rounding error when a
    // ray runs closer than asin(kappa / sin(gamma)) to a triangle's plane (gamma = angle between the edges), and may then
    // accept a "hit" far outside the triangle: the filter may skip a leaf only for rays that keep that distance from the
    // planes of ALL its triangles. kappa: DESIGN.md section 5 derives |ray - hit| <= 33 u |o - p1| / (sin(gamma) |cos(phi)|)
    // (u = 2^-24) for an accepted hit; with the filter's pad of SRX_FILTER_PAD_REL x the coordinate distance from the origin
    // to the leaf's box (>= |o - p1| / sqrt 3) that is kappa >= 0.0069, taken as 1/80.
    auto cone_word = [&](uint32_t first, uint32_t count) -> uint32_t {
        const double kappa = 1.0 / 80.0;
        struct V { double x, y, z; };
        std::vector<V> nh; std::vector<double> tau;
        V sum = {0, 0, 0};
        for (uint32_t p = first; p < first + count; p++) {
            const uint8_t* pk = rp + (size_t)p * SRX_REF_PACKET_BYTES;
            const int size = (int)(int8_t)pk[420];
            for (int k = 0; k < 8 && k < size; k++) {
                const double ax = rdf(pk, 96 + 4 * k), ay = rdf(pk, 128 + 4 * k), az = rdf(pk, 160 + 4 * k);
                const double bx = rdf(pk, 192 + 4 * k), by = rdf(pk, 224 + 4 * k), bz = rdf(pk, 256 + 4 * k);
                const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
                const double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
                const double ll = std::sqrt(ax * ax + ay * ay + az * az) * std::sqrt(bx * bx + by * by + bz * bz);
                if (!(ll >= 1e-16) || !(ll <= 1e16) || !(nn > 0.0)) return SRX_CONE_NEVER;
                const double t = kappa / (nn / ll);
                if (!(t < 1.0)) return SRX_CONE_NEVER;
                V n = {nx / nn, ny / nn, nz / nn};
                if (!nh.empty() && n.x * nh[0].x + n.y * nh[0].y + n.z * nh[0].z < 0.0) { n.x = -n.x; n.y = -n.y; n.z = -n.z; }
                nh.push_back(n); tau.push_back(t);
                sum.x += n.x; sum.y += n.y; sum.z += n.z;
            }
        }
        if (nh.empty()) return 0u;
        const double m = std::max(std::fabs(sum.x), std::max(std::fabs(sum.y), std::fabs(sum.z)));
        if (!(m > 1e-6)) return SRX_CONE_NEVER;
        const double a[3] = {std::floor(sum.x / m * 127.0 + 0.5), std::floor(sum.y / m * 127.0 + 0.5), std::floor(sum.z / m * 127.0 + 0.5)};
        const double A = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        double thr = 0.0;
        for (size_t k = 0; k < nh.size(); k++) {
            const double c = std::fabs(nh[k].x * a[0] + nh[k].y * a[1] + nh[k].z * a[2]) / A;
            if (!(tau[k] < c)) return SRX_CONE_NEVER;
            const double sn = std::sqrt(std::max(0.0, 1.0 - c * c));
            thr = std::max(thr, tau[k] * c + std::sqrt(1.0 - tau[k] * tau[k]) * sn);
        }
        // the kernel tests |a . dq| >= 127 h with dq = the ray's unit direction x 127 rounded to int8: |a . dq - 127 a . d^|
        // <= (|ax| + |ay| + |az|) / 2, hence the second term
        const double h = std::ceil(thr * A + (std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2])) / 254.0 + 0.01);
        if (!(h <= 254.0)) return SRX_CONE_NEVER;
        return ((uint32_t)(int)a[0] & 255u) | (((uint32_t)(int)a[1] & 255u) << 8) | (((uint32_t)(int)a[2] & 255u) << 16) | ((uint32_t)(int)h << 24);
    };
    float scene_scale = 1e-3f;
    if (desc->n_packets) {
        float mn[3], mx[3];
        packet_bounds(0, desc->n_packets, mn, mx);
        for (int ax = 0; ax < 3; ax++)
            if (mx[ax] >= mn[ax]) scene_scale = std::max(scene_scale, std::max(mx[ax] - mn[ax], std::max(std::fabs(mn[ax]), std::fabs(mx[ax]))));
    }
    s->g.filter_pad = SRX_FILTER_PAD_ABS * scene_scale;
    // SRX_LEAF_FILTER=0: no leaf is ever skipped (an infinite pad), i.e. the reference's walk packet for packet; for A/B
    // runs of the filter (tests/test_leaf_filter_gpu.py)
    if (const char* e = getenv("SRX_LEAF_FILTER")) if (atoi(e) == 0) s->g.filter_pad = __builtin_inff();

    // ---- nodes: 288-B SoA record (BVH.h:874-881) -> two 128-B child-major lines
    std::vector<GNode> nodes(desc->n_nodes);
    for (uint32_t i = 0; i < desc->n_nodes; i++) {
        const uint8_t* nd = rn + (size_t)i * SRX_REF_NODE_BYTES;
        for (int k = 0; k < 8; k++) {
            const uint32_t leaf = rdu(nd, 192 + 4 * k), idx = rdu(nd, 224 + 4 * k);
            if ((leaf == 0 && idx >= desc->n_nodes) || (leaf > 0 && ((uint64_t)idx + leaf > desc->n_packets))) {
                srx_scene_destroy(s); SRX_FAIL("srx_scene_create: node child index out of range");
            }
            // children are appended after their parent (BVH.h:1008-1072): anything else is not a tree this traversal ends on
            if (leaf == 0 && idx != 0 && idx <= i) { srx_scene_destroy(s); SRX_FAIL("srx_scene_create: a node's child must have a larger index than the node"); }
            uint32_t link = 0;                                        // unused slot (BVH.h:890-895)
            if (leaf == 0 && idx != 0) link = 2u * idx;
            if (leaf > 0) {
                if (leaf - 1u > SRX_LINK_MORE_MAX) { srx_scene_destroy(s); SRX_FAIL("srx_scene_create: a leaf with more than 32 packets (256 triangles) is not supported"); }
                link = SRX_LINK_LEAF | ((leaf - 1u) << SRX_LINK_MORE_SHIFT) | (2u * desc->n_nodes + 3u * idx);
            }
            float rmin[3] = {rdf(nd, 96 + 4 * k), rdf(nd, 128 + 4 * k), rdf(nd, 160 + 4 * k)};
            float rmax[3] = {rdf(nd, 0 + 4 * k), rdf(nd, 32 + 4 * k), rdf(nd, 64 + 4 * k)};
            float smin[3] = {rmin[0], rmin[1], rmin[2]}, smax[3] = {rmax[0], rmax[1], rmax[2]};
            // A leaf box that contains the origin on every axis (all ordinary leaves, quirk Q1): a reference plane at
            // 0 hides the tight plane, which is taken from the packets (to an ulp: it only feeds the padded filter and
            // an ordering the clamp keeps); a non-zero reference plane IS the tight plane.
            bool ext = leaf > 0;
            for (int ax = 0; ax < 3 && ext; ax++) ext = (rmin[ax] <= 0.0f) && (rmax[ax] >= 0.0f);
            uint32_t cone = 0u;
            if (ext) {
                cone = cone_word(idx, leaf);
                float pmn[3], pmx[3];
                packet_bounds(idx, leaf, pmn, pmx);
                for (int ax = 0; ax < 3; ax++) {
                    if (rmin[ax] == 0.0f && rmax[ax] == 0.0f) { smin[ax] = 0.0f; smax[ax] = 0.0f; }
                    else if (rmin[ax] == 0.0f) smin[ax] = std::min(std::max(pmn[ax], 0.0f), rmax[ax]);
                    else if (rmax[ax] == 0.0f) smax[ax] = std::max(std::min(pmx[ax], 0.0f), rmin[ax]);
                }
                link |= SRX_LINK_EXT;
            }
            nodes[i].a[k] = make_float4(smin[0], smin[1], smin[2], smax[0]);
            nodes[i].b[k] = make_float4(smax[1], smax[2], u2f_host(link), u2f_host(cone));
        }
    }
    // ---- packets: 448-B record (Primitives.h:127-134) -> three 128-B lanes-major lines
    std::vector<GPacket> packets(desc->n_packets);
    for (uint32_t i = 0; i < desc->n_packets; i++) {
        const uint8_t* pk = rp + (size_t)i * SRX_REF_PACKET_BYTES;
        const int32_t index = (int32_t)rdu(pk, 416);
        // what a hit hands to the shading kernel must exist (empty lanes are never hit: det = 0, Primitives.h:159-202)
        const int size = std::min(8, (int)(int8_t)pk[420]);
        for (int k = 0; k < size; k++) {
            const uint32_t mat = rdu(pk, 384 + 4 * k);
            if ((desc->n_materials > 0 && mat >= desc->n_materials) || (desc->n_tris > 0 && (index < 0 || (uint64_t)index + k >= desc->n_tris))) {
                srx_scene_destroy(s); SRX_FAIL("srx_scene_create: packet material or triangle index out of range");
            }
        }
        for (int k = 0; k < 8; k++) {
            const float p1x = rdf(pk, 0 + 4 * k), p1y = rdf(pk, 32 + 4 * k), p1z = rdf(pk, 64 + 4 * k);
            const float e1x = rdf(pk, 96 + 4 * k), e1y = rdf(pk, 128 + 4 * k), e1z = rdf(pk, 160 + 4 * k);
            const float e2x = rdf(pk, 192 + 4 * k), e2y = rdf(pk, 224 + 4 * k), e2z = rdf(pk, 256 + 4 * k);
            packets[i].a[k] = make_float4(p1x, p1y, p1z, e1x);
            packets[i].b[k] = make_float4(e1y, e1z, e2x, 0.0f);
            packets[i].c[k] = make_float2(e2y, e2z); packets[i].material[k] = rdu(pk, 384 + 4 * k);
            packets[i].tri_index[k] = (uint32_t)(index + k);
        }
    }
    // ---- shading triangles: the 160 bytes of PackagedTri (Primitives.h:62-81) that process_ray reads
    std::vector<GTri> tris(desc->n_tris);
    for (uint32_t i = 0; i < desc->n_tris; i++) {
        const uint8_t* t = rt + (size_t)i * SRX_REF_TRI_BYTES;
        float f[40];
        int w = 0;
        for (int k = 0; k < 6; k++) f[w++] = rdf(t, 36 + 4 * k);     // UV_basis, U_delta, V_delta
        f[w++] = rdf(t, 60);                                         // material (bits)
        for (int k = 0; k < 33; k++) f[w++] = rdf(t, 64 + 4 * k);    // normal, origin_offset, n1..3, t1..3, b1..3
        std::memcpy(&tris[i], f, sizeof(GTri));
        const uint32_t mat = rdu(t, 60);
        if (mat != 0xFFFFFFFFu && mat >= desc->n_materials && desc->n_materials > 0) {
            srx_scene_destroy(s); SRX_FAIL("srx_scene_create: triangle material index out of range");
        }
    }
    {   // one allocation: node lines, then packet lines
        const size_t nb = nodes.size() * sizeof(GNode), pb = packets.size() * sizeof(GPacket);
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, std::max<size_t>(nb + pb, 128));
        if (e == hipSuccess) s->allocations.push_back(d);
        if (e == hipSuccess && nb) e = hipMemcpy(d, nodes.data(), nb, hipMemcpyHostToDevice);
        if (e == hipSuccess && pb) e = hipMemcpy((char*)d + nb, packets.data(), pb, hipMemcpyHostToDevice);
        if (e != hipSuccess) { srx_scene_destroy(s); SRX_FAIL(std::string("srx_scene_create: geometry upload: ") + hipGetErrorString(e)); }
        g.geom = (const char*)d;
    }
    if (upload(s, tris.data(), tris.size(), &g.tris)) { srx_scene_destroy(s); return 1; }
    g.n_nodes = desc->n_nodes; g.n_packets = desc->n_packets; g.n_tris = desc->n_tris;

    // ---- textures (Texture.h:227-230: U_multiplier = resolution * scale, int*float -> float -> double)
    std::vector<GTexture> tex(desc->n_textures);
    for (uint32_t i = 0; i < desc->n_textures; i++) {
        const srx_texture& t = desc->textures[i];
        if (!t.rgb8 || t.width <= 0 || t.height <= 0) { srx_scene_destroy(s); SRX_FAIL("srx_scene_create: bad texture"); }
        // RGB8 rows -> RGBX8 in 32x32 tiles, Z-order inside a tile (texels outside the image are never addressed)
        const uint32_t tiles_x = ((uint32_t)t.width + 31u) / 32u, tiles_y = ((uint32_t)t.height + 31u) / 32u;
        std::vector<uint32_t> tiled((size_t)tiles_x * tiles_y * 1024u, 0u);
        auto spread5 = [](uint32_t v) { v = (v | (v << 4)) & 0x10Fu; v = (v | (v << 2)) & 0x133u; return (v | (v << 1)) & 0x155u; };
        for (uint32_t y = 0; y < (uint32_t)t.height; y++) {
            for (uint32_t x = 0; x < (uint32_t)t.width; x++) {
                const uint8_t* px = t.rgb8 + ((size_t)y * t.width + x) * 3;
                const size_t tile = (size_t)(y >> 5) * tiles_x + (x >> 5);
                tiled[tile * 1024u + (spread5(x & 31u) | (spread5(y & 31u) << 1))] = px[0] | (px[1] << 8) | (px[2] << 16);
            }
        }
        const uint32_t* d = nullptr;
        if (upload(s, tiled.data(), tiled.size(), &d)) { srx_scene_destroy(s); return 1; }
        tex[i].data = d; tex[i].res_x = t.width; tex[i].res_y = t.height;
        tex[i].U_multiplier = (double)(float)(t.width * t.scale_x);
        tex[i].V_multiplier = (double)(float)(t.height * t.scale_y);
        tex[i].normal_transform = t.normal_transform; tex[i].tiles_x = (int)tiles_x;
    }
    if (upload(s, tex.data(), tex.size(), &g.textures)) { srx_scene_destroy(s); return 1; }
    g.n_textures = desc->n_textures;

    static_assert(sizeof(GMaterial) == sizeof(srx_material), "material layout");
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        const srx_material& m = desc->materials[i];
        const int ids[8] = {m.color_tex, m.roughness_tex, m.metallic_tex, m.specular_tex, m.emissive_tex, m.normal_tex, m.transmission_tex, m.ior_tex};
        for (int id : ids) if (id >= (int)desc->n_textures) { srx_scene_destroy(s); SRX_FAIL("srx_scene_create: material texture id out of range"); }
    }
    if (upload(s, (const GMaterial*)desc->materials, desc->n_materials, &g.materials)) { srx_scene_destroy(s); return 1; }
    g.n_materials = desc->n_materials;
    static_assert(sizeof(GSun) == sizeof(srx_sun), "sun layout");
    if (desc->n_suns > 255) { srx_scene_destroy(s); SRX_FAIL("srx_scene_create: at most 255 sun lights"); }
    if (upload(s, (const GSun*)desc->suns, desc->n_suns, &g.suns)) { srx_scene_destroy(s); return 1; }
    g.n_suns = desc->n_suns;

    // ---- lookup tables
    std::vector<float> arccos, pdf;
    const float* pa = desc->arccos_lookup; const float* pp = desc->pdf_lookup;
    if (!pa || !pp) {
        arccos.resize(2048); pdf.resize(1024 * 1024);
        if (srx_build_tables(arccos.data(), pdf.data())) { srx_scene_destroy(s); return 1; }
        if (!pa) pa = arccos.data();
        if (!pp) pp = pdf.data();
    }
    if (upload(s, pa, 2048, &g.arccos_lookup)) { srx_scene_destroy(s); return 1; }
    if (upload(s, pp, (size_t)1024 * 1024, &g.pdf_lookup)) { srx_scene_destroy(s); return 1; }

    g.max_generations = desc->max_generations;
    for (int gen = 0; gen < 32; gen++) {   // Core.h:428: int = short * pow(float, char) in double
        double v = (double)(short)desc->monte_carlo_max * std::pow((double)desc->monte_carlo_modifier, (double)gen);
        g.bounce_count[gen] = (int)v;
    }
    const int bound = stack_bound(rn, desc->n_nodes);
    if (bound > 256) { srx_scene_destroy(s); SRX_FAIL("srx_scene_create: tree needs a traversal stack deeper than the reference's 256 entries (Scene.h:181)"); }
    g.stack_depth = std::max(bound, 2);
    if (g.stack_depth > SRX_SPILL_MIN_LDS_DEPTH) {
        // overflow areas of the traversal stack (srx_trace.hip): used by launches that keep fewer than `bound` entries
        // per octet in LDS
        const size_t area_bytes = (size_t)SRX_SPILL_SLOTS * 8 * (size_t)g.stack_depth * sizeof(unsigned long long);
        const size_t ctl_bytes = (size_t)(16 + SRX_SPILL_SLOTS) * sizeof(uint32_t);
        void *area = nullptr, *ctl = nullptr;
        hipError_t e = hipMalloc(&area, area_bytes);
        if (e == hipSuccess) { s->allocations.push_back(area); e = hipMalloc(&ctl, ctl_bytes); }
        if (e == hipSuccess) { s->allocations.push_back(ctl); e = hipMemset(ctl, 0, ctl_bytes); }
        if (e != hipSuccess) { srx_scene_destroy(s); SRX_FAIL(std::string("srx_scene_create: stack overflow areas: ") + hipGetErrorString(e)); }
        g.spill_area = (unsigned long long*)area; g.spill_ctl = (uint32_t*)ctl;
    }
    *out = s;
    return 0;
}

extern "C" void srx_scene_destroy(srx_scene* scene) {
    if (!scene) return;
    for (void* p : scene->allocations) (void)hipFree(p);
    delete scene;
}

extern "C" int srx_scene_stack_bound(const srx_scene* scene) { return scene ? scene->g.stack_depth : -1; }

// Called where a call has waited for its traversal kernels anyway: did a wave find no overflow area for its stack?
int srx_scene_check_spill(const srx_scene* scene) {
    if (!scene->g.spill_ctl) return 0;
    uint32_t flag = 0;
    SRX_HIP(hipMemcpy(&flag, scene->g.spill_ctl + 1, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) SRX_FAIL("srx: a traversal wave found no free stack overflow area; results of this call are incomplete");
    return 0;
}

extern "C" int64_t srx_scene_stack_spills(const srx_scene* scene) {
    if (!scene) return -1;
    if (!scene->g.spill_ctl) return 0;
    uint32_t n = 0;
    if (hipMemcpy(&n, scene->g.spill_ctl + 2, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)n;
}

// ---------------------------------------------------------------- casts
extern "C" int srx_cast_closest_device(srx_scene* scene, const float* d_org3, const float* d_dir3, uint32_t n, srx_hit* d_hits,
                                       uint64_t* d_counters2, void* stream) {
    if (!scene) SRX_FAIL("srx_cast_closest_device: null scene");
    TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.org = d_org3; a.org_stride = 3; a.dir = d_dir3; a.dir_stride = 3; a.n = n;
    a.hits = (GHit*)d_hits;
    a.counters = (unsigned long long*)d_counters2;
    SRX_HIP(srx_launch_trace(0, scene->g, a, (hipStream_t)stream));
    return 0;
}

// Folds the replicated counter slots of the static scratch into the caller's two counters.
__global__ void k_fold_static_counters(const unsigned long long* slots, unsigned long long* out2) {
    unsigned long long a = slots[2 * threadIdx.x], b = slots[2 * threadIdx.x + 1];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out2[0], a); atomicAdd(&out2[1], b); }
}

// srx_cast_closest_device for callers that cannot allocate (include/srx.h): same kernels; the traversal stack's overflow
// areas, their control words and the replicated counter slots live in this library's static device scratch instead of
// memory allocated with the scene. No hipMalloc / hipFree / allocating call on this path.
extern "C" int srx_cast_closest_device_static(srx_scene* scene, const float* d_org3, const float* d_dir3, uint32_t n, srx_hit* d_hits,
                                              uint64_t* d_counters2, void* stream) {
    if (!scene) SRX_FAIL("srx_cast_closest_device_static: null scene");
    if (scene->g.stack_depth > SRX_STATIC_MAX_STACK)
        SRX_FAIL("srx_cast_closest_device_static: the scene's traversal stack bound (" + std::to_string(scene->g.stack_depth) +
                 ") exceeds the " + std::to_string(SRX_STATIC_MAX_STACK) + " entries per ray the static scratch holds; use srx_cast_closest_device");
    int device = -1;
    SRX_HIP(hipGetDevice(&device));
    if (device != scene->device) SRX_FAIL("srx_cast_closest_device_static: the scene lives on another device than the current one");
    StaticScratch* sp = nullptr;
    if (static_scratch_for(device, &sp)) return 1;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // what must start at zero: claim cursor, error flag, spill count and the "area taken" words (a wave releases its
    // area when it ends, but a previous call may have been abandoned), and the counter slots
    SRX_HIP(hipMemsetAsync(sp->ctl, 0, sizeof(sp->ctl), st));
    if (d_counters2) SRX_HIP(hipMemsetAsync(sp->counters, 0, sizeof(sp->counters), st));
    GScene g = scene->g;
    g.spill_area = sp->area;
    g.spill_ctl = sp->ctl;
    TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.org = d_org3; a.org_stride = 3; a.dir = d_dir3; a.dir_stride = 3; a.n = n;
    a.hits = (GHit*)d_hits;
    if (d_counters2) { a.counters = sp->counters; a.counter_slot_mask = SRX_STATIC_COUNTER_SLOTS - 1; a.counter_slot_stride = 2; }
    SRX_HIP(srx_launch_trace(0, g, a, st));
    if (d_counters2) {
        hipLaunchKernelGGL(k_fold_static_counters, dim3(1), dim3(SRX_STATIC_COUNTER_SLOTS), 0, st, sp->counters, (unsigned long long*)d_counters2);
        SRX_HIP(hipGetLastError());
    }
    return 0;
}

// Did a wave of a srx_cast_closest_device_static call on the CURRENT device find no free overflow area (its rays then
// ended early: hits or misses that are not the reference's)? Cannot happen while the calls on a GPU do not overlap (there
// are more areas than wave slots); two overlapping calls share the areas and re-zero each other's control words, and this
// flag is how a caller finds out. Waits for `stream`, reads one word (no allocation), returns 0 with *truncated set.
extern "C" int srx_static_scratch_status(void* stream, int32_t* truncated) {
    if (!truncated) SRX_FAIL("srx_static_scratch_status: null argument");
    int device = -1;
    SRX_HIP(hipGetDevice(&device));
    StaticScratch* sp = nullptr;
    if (static_scratch_for(device, &sp)) return 1;
    SRX_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint32_t flag = 0;
    SRX_HIP(hipMemcpy(&flag, &sp->ctl[1], sizeof(flag), hipMemcpyDeviceToHost));
    *truncated = flag ? 1 : 0;
    return 0;
}

extern "C" int srx_cast_shadow_device(srx_scene* scene, const float* d_org3, const float* d_dir3, uint32_t n, uint8_t* d_lit, void* stream) {
    if (!scene) SRX_FAIL("srx_cast_shadow_device: null scene");
    TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.org = d_org3; a.org_stride = 3; a.dir = d_dir3; a.dir_stride = 3; a.n = n;
    a.lit = d_lit;
    SRX_HIP(srx_launch_trace(1, scene->g, a, (hipStream_t)stream));
    return 0;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" int srx_cast_closest(srx_scene* scene, const float* org3, const float* dir3, uint32_t n, srx_hit* hits,
                                float* advanced_org3, uint64_t* counters2) {
    if (!scene) SRX_FAIL("srx_cast_closest: null scene");
    if (n == 0) { if (counters2) counters2[0] = counters2[1] = 0; return 0; }
    DevBuf o, d, h, adv, c;
    SRX_HIP(hipMalloc(&o.p, (size_t)n * 12)); SRX_HIP(hipMalloc(&d.p, (size_t)n * 12));
    SRX_HIP(hipMalloc(&h.p, (size_t)n * sizeof(srx_hit)));
    SRX_HIP(hipMalloc(&c.p, 16));
    if (advanced_org3) SRX_HIP(hipMalloc(&adv.p, (size_t)n * 12));
    SRX_HIP(hipMemcpy(o.p, org3, (size_t)n * 12, hipMemcpyHostToDevice));
    SRX_HIP(hipMemcpy(d.p, dir3, (size_t)n * 12, hipMemcpyHostToDevice));
    SRX_HIP(hipMemset(c.p, 0, 16));
    TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.org = (const float*)o.p; a.org_stride = 3; a.dir = (const float*)d.p; a.dir_stride = 3; a.n = n;
    a.hits = (GHit*)h.p; a.advanced_org3 = (float*)adv.p; a.counters = (unsigned long long*)c.p;
    SRX_HIP(srx_launch_trace(0, scene->g, a, 0));
    SRX_HIP(hipDeviceSynchronize());
    SRX_HIP(hipMemcpy(hits, h.p, (size_t)n * sizeof(srx_hit), hipMemcpyDeviceToHost));
    if (advanced_org3) SRX_HIP(hipMemcpy(advanced_org3, adv.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    if (counters2) SRX_HIP(hipMemcpy(counters2, c.p, 16, hipMemcpyDeviceToHost));
    return srx_scene_check_spill(scene);
}

extern "C" int srx_cast_shadow(srx_scene* scene, const float* org3, const float* dir3, uint32_t n, uint8_t* lit) {
    if (!scene) SRX_FAIL("srx_cast_shadow: null scene");
    if (n == 0) return 0;
    DevBuf o, d, l;
    SRX_HIP(hipMalloc(&o.p, (size_t)n * 12)); SRX_HIP(hipMalloc(&d.p, (size_t)n * 12)); SRX_HIP(hipMalloc(&l.p, n));
    SRX_HIP(hipMemcpy(o.p, org3, (size_t)n * 12, hipMemcpyHostToDevice));
    SRX_HIP(hipMemcpy(d.p, dir3, (size_t)n * 12, hipMemcpyHostToDevice));
    TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.org = (const float*)o.p; a.org_stride = 3; a.dir = (const float*)d.p; a.dir_stride = 3; a.n = n;
    a.lit = (uint8_t*)l.p;
    SRX_HIP(srx_launch_trace(1, scene->g, a, 0));
    SRX_HIP(hipDeviceSynchronize());
    SRX_HIP(hipMemcpy(lit, l.p, n, hipMemcpyDeviceToHost));
    return srx_scene_check_spill(scene);
}
