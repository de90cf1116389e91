// oracle/sro_capi.cpp — TEST INFRASTRUCTURE ONLY (the parity oracle).
//
// extern "C" surface over the CPU restatement so tests/, smoke() and
// bench.py's cpu_baseline leg can drive it through ctypes. The product never
// links or loads this library. PARITY UNPINNED (see sro_math.h).
#include "sro_shade.h"
#include "sro_noise.h"
#include "sro_post.h"
#include <chrono>
#include <thread>
#include <atomic>
#include <string>

using namespace sro;

namespace {
struct Handle {
    Scene scene;
    std::vector<std::vector<uint8_t>> tex_store;
    std::string err;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

// Authoring triangle as handed over by the host (world space, already flattened).
struct sro_tri_in {
    float p[9];        // p1 p2 p3
    float uv[6];       // UV_1 UV_2 UV_3
    float n[9];        // vertex normals (used when flags&1, else the face normal — Primitives.h:238-252)
    int32_t material;
    uint32_t flags;
};

struct sro_material_in {           // mirrors sro::Material field order
    float color[3];     int32_t color_tex;
    float roughness;    int32_t roughness_tex;
    float metallic;     int32_t metallic_tex;
    float specular;     int32_t specular_tex;
    float emissive[3];  int32_t emissive_tex;
    float normal[3];    int32_t normal_tex;
    float transmission; int32_t transmission_tex;
    float ior;          int32_t ior_tex;
    int32_t use_normals;
};

struct sro_hit { uint32_t material; float t, u, v; int32_t tri_index; };

struct sro_camera_in {
    float position[3]; float rotation[4]; float focal_position[3];
    float lens_width, lens_height;
    int32_t res_x, res_y, subdiv;
};

const char* sro_last_error() { return g_err.c_str(); }

void* sro_scene_new() { return new Handle(); }
void sro_scene_free(void* h) { delete (Handle*)h; }

int sro_scene_package(void* hv, const sro_tri_in* tris, uint32_t n) {
    Handle* h = (Handle*)hv;
    try {
        std::vector<Tri> v;
        v.reserve(n);
        for (uint32_t i = 0; i < n; i++) {
            const sro_tri_in& t = tris[i];
            Tri T(XYZ(t.p[0], t.p[1], t.p[2]), XYZ(t.p[3], t.p[4], t.p[5]), XYZ(t.p[6], t.p[7], t.p[8]),
                  XY(t.uv[0], t.uv[1]), XY(t.uv[2], t.uv[3]), XY(t.uv[4], t.uv[5]), (short)t.material);
            if (t.flags & 1u) {
                T.n1 = XYZ(t.n[0], t.n[1], t.n[2]); T.n2 = XYZ(t.n[3], t.n[4], t.n[5]); T.n3 = XYZ(t.n[6], t.n[7], t.n[8]);
            }
            v.push_back(T);
        }
        h->scene.geo = PackagedGeometry();
        auto t0 = std::chrono::steady_clock::now();
        package(v, h->scene.geo);
        h->scene.geo.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        h->scene.geo.build_filter_boxes();
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// Install already-packaged geometry (reference layouts) instead of building it.
int sro_scene_set_packaged(void* hv, const void* nodes, uint32_t n_nodes, const void* packets, uint32_t n_packets,
                           const void* tris, uint32_t n_tris) {
    Handle* h = (Handle*)hv;
    PackagedGeometry& g = h->scene.geo;
    g.avx_bvh.assign((const BVH_AVX*)nodes, (const BVH_AVX*)nodes + n_nodes);
    g.avx_tri_data.assign((const PTri_AVX*)packets, (const PTri_AVX*)packets + n_packets);
    g.tri_data.assign((const PackagedTri*)tris, (const PackagedTri*)tris + n_tris);
    g.build_filter_boxes();
    return 0;
}

// mirrors the product's SRX_LEAF_FILTER=0: with the filter off nothing is counted as saved
void sro_scene_set_leaf_filter(void* hv, int32_t enabled) { ((Handle*)hv)->scene.geo.filter_enabled = enabled != 0; }

// 1 = casts (and everything built on them) use the AVX2-intrinsic node and packet tests of sro_trace_avx2.h: same results
// bit for bit, no leaf-filter bookkeeping in the counters; what bench.py times as the CPU baseline
void sro_scene_set_simd(void* hv, int32_t enabled) { ((Handle*)hv)->scene.geo.simd = enabled != 0; }

uint32_t sro_scene_counts(void* hv, uint32_t* n_nodes, uint32_t* n_packets, uint32_t* n_tris) {
    Handle* h = (Handle*)hv;
    *n_nodes = (uint32_t)h->scene.geo.avx_bvh.size();
    *n_packets = (uint32_t)h->scene.geo.avx_tri_data.size();
    *n_tris = (uint32_t)h->scene.geo.tri_data.size();
    return 0;
}
const void* sro_scene_nodes(void* hv) { return ((Handle*)hv)->scene.geo.avx_bvh.data(); }
const void* sro_scene_packets(void* hv) { return ((Handle*)hv)->scene.geo.avx_tri_data.data(); }
const void* sro_scene_tris(void* hv) { return ((Handle*)hv)->scene.geo.tri_data.data(); }
double sro_scene_build_seconds(void* hv) { return ((Handle*)hv)->scene.geo.build_seconds; }

int sro_scene_set_materials(void* hv, const sro_material_in* m, uint32_t n) {
    Handle* h = (Handle*)hv;
    h->scene.materials.clear();
    for (uint32_t i = 0; i < n; i++) {
        Material M;
        M.color = XYZ(m[i].color[0], m[i].color[1], m[i].color[2]); M.color_tex = m[i].color_tex;
        M.roughness = m[i].roughness; M.roughness_tex = m[i].roughness_tex;
        M.metallic = m[i].metallic; M.metallic_tex = m[i].metallic_tex;
        M.specular = m[i].specular; M.specular_tex = m[i].specular_tex;
        M.emissive = XYZ(m[i].emissive[0], m[i].emissive[1], m[i].emissive[2]); M.emissive_tex = m[i].emissive_tex;
        M.normal = XYZ(m[i].normal[0], m[i].normal[1], m[i].normal[2]); M.normal_tex = m[i].normal_tex;
        M.transmission = m[i].transmission; M.transmission_tex = m[i].transmission_tex;
        M.IOR = m[i].ior; M.IOR_tex = m[i].ior_tex;
        M.use_normals = m[i].use_normals != 0;
        h->scene.materials.push_back(M);
    }
    return 0;
}

int sro_scene_add_texture(void* hv, const uint8_t* rgb8, int32_t w, int32_t hgt, float sx, float sy, int32_t normal_transform) {
    Handle* h = (Handle*)hv;
    h->tex_store.emplace_back(rgb8, rgb8 + (size_t)w * hgt * 3);
    Texture T;
    T.resolution_x = w; T.resolution_y = hgt; T.scale = XY(sx, sy);
    T.use_normal_transform = normal_transform != 0;
    T.prep();
    h->scene.textures.push_back(T);
    for (size_t i = 0; i < h->scene.textures.size(); i++) h->scene.textures[i].data = h->tex_store[i].data();
    return (int)h->scene.textures.size() - 1;
}

int sro_scene_add_sun(void* hv, const float* dir3, const float* emission3) {
    Handle* h = (Handle*)hv;
    SunLight L;
    L.rotation = XYZ(dir3[0], dir3[1], dir3[2]);
    L.emission = XYZ(emission3[0], emission3[1], emission3[2]);
    h->scene.lights.push_back(L);
    return 0;
}

int sro_scene_set_params(void* hv, int32_t max_generations, int32_t monte_carlo_max, float monte_carlo_modifier) {
    Handle* h = (Handle*)hv;
    h->scene.max_generations = (short)max_generations;
    h->scene.monte_carlo_max = (short)monte_carlo_max;
    h->scene.monte_carlo_modifier = monte_carlo_modifier;
    return 0;
}

// Build (or install) the two lookup tables: arccos (2048) and the NDF inverse CDF (1024x1024).
int sro_scene_build_tables(void* hv) {
    Handle* h = (Handle*)hv;
    build_arccos(h->scene.tables.arccos_lookup);
    pdf::prep(h->scene.tables.pdf_lookup);
    return 0;
}
int sro_scene_set_tables(void* hv, const float* arccos2048, const float* pdf1m) {
    Handle* h = (Handle*)hv;
    h->scene.tables.arccos_lookup.assign(arccos2048, arccos2048 + 2048);
    h->scene.tables.pdf_lookup.assign(pdf1m, pdf1m + 1024 * 1024);
    return 0;
}
const float* sro_scene_arccos(void* hv) { return ((Handle*)hv)->scene.tables.arccos_lookup.data(); }
const float* sro_scene_pdf(void* hv) { return ((Handle*)hv)->scene.tables.pdf_lookup.data(); }

// ---- casts ---------------------------------------------------------------
// counters6 (optional) = {node visits, packet tests, max stack depth, packet tests the product's leaf filter saves,
// closest-hit updates taken inside a leaf that filter skips (any > 0 = the filter changes a result), lanes that pass the
// triangle test with t > 0 in such a leaf, packet tests in leaves the box test would skip but the cone guard keeps}.
int sro_cast_closest(void* hv, const float* org3, const float* dir3, uint32_t n, sro_hit* out, float* advanced_org3,
                     uint64_t* counters4) {
    Handle* h = (Handle*)hv;
    TraceCounters c;
    try {
        for (uint32_t i = 0; i < n; i++) {
            XYZ o(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]);
            XYZ d(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]);
            CastResults r = execute_ray_cast(h->scene.geo, o, d, &c);
            out[i].material = r.material;
            out[i].t = r.TUV.X; out[i].u = r.TUV.Y; out[i].v = r.TUV.Z;
            out[i].tri_index = r.tri_index;
            if (advanced_org3) { advanced_org3[3 * i] = o.X; advanced_org3[3 * i + 1] = o.Y; advanced_org3[3 * i + 2] = o.Z; }
        }
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    if (counters4) {
        counters4[0] = c.nodes; counters4[1] = c.packets; counters4[2] = (uint64_t)c.max_stack; counters4[3] = c.packets_unhittable;
        counters4[4] = c.unhittable_accepts; counters4[5] = c.unhittable_passes; counters4[6] = c.packets_guarded;
    }
    return 0;
}

// The leaf filter's observed margins over a ray set (TraceCounters::max_r_over_T, max_outside_over_D, margin_lanes): scalar walk.
int sro_filter_margins(void* hv, const float* org3, const float* dir3, uint32_t n, double* out3) {
    Handle* h = (Handle*)hv;
    TraceCounters c;
    const bool simd = h->scene.geo.simd;
    h->scene.geo.simd = false;
    try {
        for (uint32_t i = 0; i < n; i++) {
            XYZ o(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]);
            XYZ d(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]);
            (void)execute_ray_cast(h->scene.geo, o, d, &c);
        }
    } catch (const std::exception& e) { h->scene.geo.simd = simd; g_err = e.what(); return 1; }
    h->scene.geo.simd = simd;
    out3[0] = c.max_r_over_T; out3[1] = c.max_outside_over_D; out3[2] = (double)c.margin_lanes;
    return 0;
}

// Shadow cast toward a sun direction: lit iff the closest hit is farther than 9999 (Primitives.h:44-50).
int sro_cast_shadow(void* hv, const float* org3, const float* dir3, uint32_t n, uint8_t* lit) {
    Handle* h = (Handle*)hv;
    try {
        for (uint32_t i = 0; i < n; i++) {
            XYZ o(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]);
            XYZ d(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]);
            CastResults r = execute_ray_cast(h->scene.geo, o, d, nullptr);
            lit[i] = r.distance > 9999 ? 1 : 0;
        }
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    return 0;
}

// ---- shading -------------------------------------------------------------
// process_ray on a batch of rays with per-ray keyed RNG (the wavefront renderer's scheme).
// stats10 (optional) = {rays_processed, diffuses_cast, shadows_cast, nodes, packets, shadow nodes, shadow packets, hits,
//                      packets / shadow packets the product's leaf filter saves}
int sro_process_rays_keyed(void* hv, const float* org3, const float* dir3, const uint64_t* keys, int32_t generation,
                           uint32_t n, float* radiance3, uint64_t* stats8) {
    Handle* h = (Handle*)hv;
    RayEngine RE; RE.data = &h->scene; RE.keyed = true;
    Diagnostics st;
    try {
        for (uint32_t i = 0; i < n; i++) {
            PackagedRay r;
            r.position = XYZ(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]);
            r.slope = XYZ(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]);
            r.generation = (signed char)generation;
            r.key = keys[i];
            XYZ L = RE.process_ray(st, r);
            radiance3[3 * i] = L.X; radiance3[3 * i + 1] = L.Y; radiance3[3 * i + 2] = L.Z;
        }
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    if (stats8) {
        stats8[0] = st.rays_processed; stats8[1] = st.diffuses_cast; stats8[2] = st.shadows_cast;
        stats8[3] = st.trace.nodes; stats8[4] = st.trace.packets;
        stats8[5] = st.shadow_trace.nodes; stats8[6] = st.shadow_trace.packets; stats8[7] = st.hits;
        stats8[8] = st.trace.packets_unhittable; stats8[9] = st.shadow_trace.packets_unhittable;
    }
    return 0;
}

static Camera make_camera(const sro_camera_in* c) {
    Camera cam;
    cam.position = XYZ(c->position[0], c->position[1], c->position[2]);
    cam.rotation.X = c->rotation[0]; cam.rotation.Y = c->rotation[1]; cam.rotation.Z = c->rotation[2]; cam.rotation.W = c->rotation[3];
    cam.focal_position = XYZ(c->focal_position[0], c->focal_position[1], c->focal_position[2]);
    cam.width = c->lens_width; cam.height = c->lens_height;
    cam.prep(c->res_x, c->res_y, c->subdiv);
    return cam;
}

// Primary ray directions for a rectangle of pixels with the keyed RNG: writes dir3 and the post-jitter key.
int sro_camera_rays_keyed(const sro_camera_in* c, int32_t x0, int32_t y0, int32_t w, int32_t hgt, uint32_t pass_index,
                          uint64_t seed, float* dir3, uint64_t* keys) {
    Camera cam = make_camera(c);
    const uint64_t npix = (uint64_t)c->res_x * c->res_y;
    size_t k = 0;
    for (int y = y0; y < y0 + hgt; y++)
        for (int x = x0; x < x0 + w; x++, k++) {
            KeyedGen g;
            g.s = KeyedGen::primary_key(seed, (uint64_t)pass_index * npix + (uint64_t)y * c->res_x + x);
            XYZ d = cam.random_slope_at(x, y, (int)(pass_index % (uint32_t)(c->subdiv * c->subdiv)), g);
            dir3[3 * k] = d.X; dir3[3 * k + 1] = d.Y; dir3[3 * k + 2] = d.Z;
            keys[k] = g.s;
        }
    return 0;
}

// Parity ray sets of fixed XorRandom seeds. Primaries: Camera::random_slope_at over the rectangle in row-major
// order with ONE sequential XorGen seeded with seed4 (two fRand draws per pixel, Camera.h:95).
int sro_camera_rays_xorgen(const sro_camera_in* c, int32_t x0, int32_t y0, int32_t w, int32_t hgt, uint32_t pass_index,
                           const uint64_t* seed4, float* dir3) {
    Camera cam = make_camera(c);
    XorGen g; g.seed(seed4[0], seed4[1], seed4[2], seed4[3]);
    size_t k = 0;
    for (int y = y0; y < y0 + hgt; y++)
        for (int x = x0; x < x0 + w; x++, k++) {
            XYZ d = cam.random_slope_at(x, y, (int)(pass_index % (uint32_t)(c->subdiv * c->subdiv)), g);
            dir3[3 * k] = d.X; dir3[3 * k + 1] = d.Y; dir3[3 * k + 2] = d.Z;
        }
    return 0;
}
// First-bounce rays: process_ray in REFERENCE mode (one sequential XorGen seeded with seed4 shared by all n primaries,
// the recursive estimator's own queue, Core.h:498-591) on generation-0 rays; the child rays each call queues
// (Core.h:573-588: position, slope) are written out, at most `capacity` of them, parent[i] = index of the primary.
// *n_children = how many were queued in all.
int sro_first_bounce_rays(void* hv, const float* org3, const float* dir3, uint32_t n, const uint64_t* seed4, uint32_t capacity,
                          float* child_org3, float* child_dir3, uint32_t* parent, uint32_t* n_children) {
    Handle* h = (Handle*)hv;
    RayEngine RE; RE.data = &h->scene; RE.keyed = false;
    XorGen xg; xg.seed(seed4[0], seed4[1], seed4[2], seed4[3]);
    RE.shared_gen = &xg;
    std::vector<PackagedRay> q;
    RE.capture_first_bounce = &q;
    Diagnostics st;
    uint32_t total = 0;
    try {
        for (uint32_t i = 0; i < n; i++) {
            PackagedRay r;
            r.position = XYZ(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]);
            r.slope = XYZ(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]);
            r.generation = 0;
            q.clear();
            (void)RE.process_ray(st, r);
            for (const PackagedRay& c : q) {
                if (total < capacity) {
                    child_org3[3 * total] = c.position.X; child_org3[3 * total + 1] = c.position.Y; child_org3[3 * total + 2] = c.position.Z;
                    child_dir3[3 * total] = c.slope.X; child_dir3[3 * total + 1] = c.slope.Y; child_dir3[3 * total + 2] = c.slope.Z;
                    parent[total] = i;
                }
                total++;
            }
        }
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    *n_children = total;
    return 0;
}

// One sample pass over a pixel rectangle, accumulated with the running-mean contract
// (SceneManager.h:204-236). accum4/aux4 are res_x*res_y float4 = (r,g,b,sample count).
//   mode 0: keyed RNG (what the GPU renderer computes; non-finite samples are skipped)
//   mode 1: reference RNG — one sequential XorGen per worker thread, non-finite samples redrawn
// threads>1 splits rows over std::thread workers (interleaved rows, like assign_iterative_groups).
// stats8 as in sro_process_rays_keyed. Returns 0 on success.
int sro_render_pass(void* hv, const sro_camera_in* c, int32_t x0, int32_t y0, int32_t w, int32_t hgt,
                    uint32_t pass_index, uint64_t seed, int32_t mode, int32_t threads, float* accum4, float* aux4,
                    uint64_t* stats8) {
    Handle* h = (Handle*)hv;
    Camera cam = make_camera(c);
    const uint64_t npix = (uint64_t)c->res_x * c->res_y;
    const int spp = c->subdiv * c->subdiv;
    if (threads < 1) threads = 1;
    std::vector<Diagnostics> st(threads);
    std::atomic<int> failed{0};
    auto worker = [&](int tid) {
        try {
            RayEngine RE; RE.data = &h->scene; RE.keyed = (mode == 0);
            XorGen xg;
            xg.seed(seed + 1 + 4ull * tid + 1000003ull * pass_index, seed + 2 + 4ull * tid, seed + 3 + 4ull * tid, seed + 4 + 4ull * tid);
            RE.shared_gen = &xg;
            for (int y = y0 + tid; y < y0 + hgt; y += threads) {
                for (int x = x0; x < x0 + w; x++) {
                    XYZ res;
                    for (;;) {
                        PackagedRay ray;
                        ray.position = cam.position;
                        ray.generation = 0;
                        if (mode == 0) {
                            KeyedGen g;
                            g.s = KeyedGen::primary_key(seed, (uint64_t)pass_index * npix + (uint64_t)y * c->res_x + x);
                            ray.slope = cam.random_slope_at(x, y, (int)(pass_index % (uint32_t)spp), g);
                            ray.key = g.s;
                        } else {
                            ray.slope = cam.random_slope_at(x, y, (int)(pass_index % (uint32_t)spp), xg);
                        }
                        res = RE.process_ray(st[tid], ray);
                        if (res.wellDefined() || mode == 0) break;     // SceneManager.h:220-223 retry
                    }
                    if (!res.wellDefined()) continue;
                    float* o = accum4 + 4 * ((size_t)y * c->res_x + x);
                    {   // SceneManager.h:224-225
                        int n = (int)o[3];
                        XYZ cur(o[0], o[1], o[2]);
                        cur *= (float)((double)n / (n + 1));
                        cur += (float)(1.0 / (n + 1)) * res;
                        o[0] = cur.X; o[1] = cur.Y; o[2] = cur.Z; o[3] = (float)(n + 1);
                    }
                    if (aux4 && (pass_index & 1u)) {   // add_aux toggles per pass starting false (:226-233)
                        float* a = aux4 + 4 * ((size_t)y * c->res_x + x);
                        int n = (int)a[3];
                        XYZ cur(a[0], a[1], a[2]);
                        cur *= (float)((double)n / (n + 1));
                        cur += (float)(1.0 / (n + 1)) * res;
                        a[0] = cur.X; a[1] = cur.Y; a[2] = cur.Z; a[3] = (float)(n + 1);
                    }
                }
            }
        } catch (const std::exception& e) { failed = 1; }
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    if (failed) { g_err = "oracle render worker failed"; return 1; }
    if (stats8) {
        for (int i = 0; i < 10; i++) stats8[i] = 0;
        for (auto& s : st) {
            stats8[0] += s.rays_processed; stats8[1] += s.diffuses_cast; stats8[2] += s.shadows_cast;
            stats8[3] += s.trace.nodes; stats8[4] += s.trace.packets;
            stats8[5] += s.shadow_trace.nodes; stats8[6] += s.shadow_trace.packets; stats8[7] += s.hits;
            stats8[8] += s.trace.packets_unhittable; stats8[9] += s.shadow_trace.packets_unhittable;
        }
    }
    return 0;
}

// ---- adaptive-sampling controller (SceneManager.h:526-692) ------------------
void sro_noise_error_map(const float* accum4, const float* aux4, uint64_t npix, float* err, int32_t* min_samples, int32_t abs_luminance) {
    int mn = npix ? (int)accum4[3] : 0;
    for (uint64_t i = 0; i < npix; i++) {
        err[i] = noise_pixel_error(accum4 + 4 * i, aux4 + 4 * i, abs_luminance != 0);
        mn = std::min(mn, (int)accum4[4 * i + 3]);
    }
    if (min_samples) *min_samples = mn;
}
void* sro_noise_new(int32_t res_x, int32_t res_y, uint64_t seed, int32_t minimum_samples) {
    NoiseController* c = new NoiseController();
    c->reset(res_x, res_y, seed);
    if (minimum_samples >= 0) c->minimum_samples = minimum_samples;
    return c;
}
void sro_noise_free(void* c) { delete (NoiseController*)c; }
void sro_noise_analyze(void* cv, const float* err, int32_t seen) { ((NoiseController*)cv)->analyze(err, seen); }
uint32_t sro_noise_region_count(void* cv) { return (uint32_t)((NoiseController*)cv)->regions.size(); }
// rects4: x0,y0,x1,y1 per region; stats5: active, noise, max_noise, min_noise, avg_noise
void sro_noise_regions(void* cv, int32_t* rects4, float* stats5) {
    NoiseController* c = (NoiseController*)cv;
    for (size_t i = 0; i < c->regions.size(); i++) {
        const NoiseRegion& r = c->regions[i];
        rects4[4 * i] = r.x0; rects4[4 * i + 1] = r.y0; rects4[4 * i + 2] = r.x1; rects4[4 * i + 3] = r.y1;
        stats5[5 * i] = r.active ? 1.0f : 0.0f; stats5[5 * i + 1] = r.noise; stats5[5 * i + 2] = r.max_noise;
        stats5[5 * i + 3] = r.min_noise; stats5[5 * i + 4] = r.avg_noise;
    }
}
int32_t sro_noise_priorities(void* cv, int32_t seen, float* out) { return ((NoiseController*)cv)->priorities(seen, out) ? 1 : 0; }
int32_t sro_noise_block_fires(float priority, uint64_t seed, uint32_t visit_index, uint32_t block, uint32_t n_blocks) {
    return noise_block_fires(priority, seed, visit_index, block, n_blocks) ? 1 : 0;
}

// One visit of every 4x4 block of the rectangle (RenderThread::process_iteratively, SceneManager.h:193-237) with
// per-block priorities (NULL = all 1): a block that fires takes one keyed sample per pixel; the stratum and the
// running-mean weight come from the pixel's own count so far (block->iterations_done), the aux buffer takes the odd
// ones (:226-233).
int sro_render_pass_adaptive(void* hv, const sro_camera_in* c, int32_t x0, int32_t y0, int32_t w, int32_t hgt,
                             uint32_t visit_index, uint64_t seed, const float* block_priority, int32_t threads,
                             float* accum4, float* aux4, uint64_t* stats8) {
    Handle* h = (Handle*)hv;
    Camera cam = make_camera(c);
    const uint64_t npix = (uint64_t)c->res_x * c->res_y;
    const int spp = c->subdiv * c->subdiv;
    const int bs = NoiseController::block_size;
    const uint32_t bxn = (uint32_t)((c->res_x + bs - 1) / bs), byn = (uint32_t)((c->res_y + bs - 1) / bs);
    if (threads < 1) threads = 1;
    std::vector<Diagnostics> st(threads);
    std::atomic<int> failed{0};
    auto worker = [&](int tid) {
        try {
            RayEngine RE; RE.data = &h->scene; RE.keyed = true;
            for (int y = y0 + tid; y < y0 + hgt; y += threads) {
                for (int x = x0; x < x0 + w; x++) {
                    const uint32_t block = (uint32_t)(y / bs) * bxn + (uint32_t)(x / bs);
                    const float pr = block_priority ? block_priority[block] : 1.0f;
                    if (!noise_block_fires(pr, seed, visit_index, block, bxn * byn)) continue;
                    float* o = accum4 + 4 * ((size_t)y * c->res_x + x);
                    const int n = (int)o[3];
                    PackagedRay ray;
                    ray.position = cam.position;
                    ray.generation = 0;
                    KeyedGen g;
                    // stratum from the pixel's count; stream keyed by the visit (a rejected sample is not repeated)
                    g.s = KeyedGen::primary_key(seed, (uint64_t)visit_index * npix + (uint64_t)y * c->res_x + x);
                    ray.slope = cam.random_slope_at(x, y, n % spp, g);
                    ray.key = g.s;
                    XYZ res = RE.process_ray(st[tid], ray);
                    if (!res.wellDefined()) continue;
                    XYZ cur(o[0], o[1], o[2]);
                    cur *= (float)((double)n / (n + 1));
                    cur += (float)(1.0 / (n + 1)) * res;
                    o[0] = cur.X; o[1] = cur.Y; o[2] = cur.Z; o[3] = (float)(n + 1);
                    if (aux4 && (n & 1)) {
                        float* a = aux4 + 4 * ((size_t)y * c->res_x + x);
                        const int na = (int)a[3];
                        XYZ ca(a[0], a[1], a[2]);
                        ca *= (float)((double)na / (na + 1));
                        ca += (float)(1.0 / (na + 1)) * res;
                        a[0] = ca.X; a[1] = ca.Y; a[2] = ca.Z; a[3] = (float)(na + 1);
                    }
                }
            }
        } catch (const std::exception& e) { failed = 1; }
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    if (failed) { g_err = "oracle render worker failed"; return 1; }
    if (stats8) {
        for (int i = 0; i < 10; i++) stats8[i] = 0;
        for (auto& s : st) {
            stats8[0] += s.rays_processed; stats8[1] += s.diffuses_cast; stats8[2] += s.shadows_cast;
            stats8[3] += s.trace.nodes; stats8[4] += s.trace.packets;
            stats8[5] += s.shadow_trace.nodes; stats8[6] += s.shadow_trace.packets; stats8[7] += s.hits;
            stats8[8] += s.trace.packets_unhittable; stats8[9] += s.shadow_trace.packets_unhittable;
        }
    }
    return 0;
}

// ---- unit hooks for fixtures / parity of the small pieces -----------------
void sro_xorgen_stream(const uint64_t* seed4, uint32_t n_scalar, uint32_t n_pools, float* scalars, float* pools8,
                       uint64_t* raw) {
    XorGen g; g.seed(seed4[0], seed4[1], seed4[2], seed4[3]);
    for (uint32_t i = 0; i < n_pools; i++) { g.regen_pool(); for (int k = 0; k < 8; k++) pools8[8 * i + k] = g.pool[k]; }
    for (uint32_t i = 0; i < n_scalar; i++) scalars[i] = g.fRand(0, 1);
    if (raw) for (int i = 0; i < 4; i++) raw[i] = g.xe_next();
}
void sro_keyed_stream(uint64_t key, uint32_t n_pools, uint32_t n_scalar, float* pools8, float* scalars, uint64_t* child_key) {
    KeyedGen g; g.s = key;
    for (uint32_t i = 0; i < n_pools; i++) { g.regen_pool(); for (int k = 0; k < 8; k++) pools8[8 * i + k] = g.pool[k]; }
    for (uint32_t i = 0; i < n_scalar; i++) scalars[i] = g.fRand(0, 1);
    if (child_key) *child_key = g.child_key();
}
uint64_t sro_primary_key(uint64_t seed, uint64_t sample_index) { return KeyedGen::primary_key(seed, sample_index); }
void sro_trig(const float* x, uint32_t n, float* s, float* c, float* t) {
    for (uint32_t i = 0; i < n; i++) { s[i] = full_sin(x[i]); c[i] = full_cos(x[i]); t[i] = vtan(x[i]); }
}
// m records of 10 fields x 8 lanes (a.xyz, b.xyz, c.xyz, s) -> m records of 5 fields x 8 lanes (cross(a,b).xyz, dot(a,b), dot_mul(a,c,s))
void sro_avx_ops(const float* in80, uint32_t m, float* out40) {
    for (uint32_t i = 0; i < m; i++) {
        const float* p = in80 + (size_t)i * 80;
        float* o = out40 + (size_t)i * 40;
        for (int l = 0; l < 8; l++) {
            cross_avx_lane(p[l], p[8 + l], p[16 + l], p[24 + l], p[32 + l], p[40 + l], o[l], o[8 + l], o[16 + l]);
            o[24 + l] = dot_avx_lane(p[l], p[8 + l], p[16 + l], p[24 + l], p[32 + l], p[40 + l]);
            o[32 + l] = dot_mul_avx_lane(p[l], p[8 + l], p[16 + l], p[48 + l], p[56 + l], p[64 + l], p[72 + l]);
        }
    }
}
void sro_poly_acos(const float* x, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = MaterialSample::poly_acos(x[i]);
}
void sro_vec_ops(const float* a3, const float* b3, uint32_t n, float* reflect3, float* cross3, float* norm3, float* dot1) {
    for (uint32_t i = 0; i < n; i++) {
        XYZ a(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
        XYZ r = XYZ::reflect(a, b), c = XYZ::cross(a, b), nn = XYZ::normalize(a);
        reflect3[3 * i] = r.X; reflect3[3 * i + 1] = r.Y; reflect3[3 * i + 2] = r.Z;
        cross3[3 * i] = c.X; cross3[3 * i + 1] = c.Y; cross3[3 * i + 2] = c.Z;
        norm3[3 * i] = nn.X; norm3[3 * i + 1] = nn.Y; norm3[3 * i + 2] = nn.Z;
        dot1[i] = XYZ::dot(a, b);
    }
}
void sro_misc(const float* a3, const float* b3, float c, uint32_t n, float* fminus3, float* mix3, float* cdot1, float* slope3) {
    for (uint32_t i = 0; i < n; i++) {
        XYZ a(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
        XYZ f = float_minus_xyz(c, a), m = XYZ::linear_mix(c, a, b), s = XYZ::slope(a, b);
        fminus3[3 * i] = f.X; fminus3[3 * i + 1] = f.Y; fminus3[3 * i + 2] = f.Z;
        mix3[3 * i] = m.X; mix3[3 * i + 1] = m.Y; mix3[3 * i + 2] = m.Z;
        cdot1[i] = XYZ::cdot(a, b);
        slope3[3 * i] = s.X; slope3[3 * i + 1] = s.Y; slope3[3 * i + 2] = s.Z;
    }
}
// Matrix3x3::applyTBNMatrix (Matrix.h:62-64) as restated in sro_math.h: pinned against the reference's header by tests
void sro_apply_tbn(const float* p3, const float* t3, const float* b3, const float* n3, uint32_t n, float* out3) {
    for (uint32_t i = 0; i < n; i++) {
        XYZ r = applyTBNMatrix(XYZ(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), XYZ(t3[3 * i], t3[3 * i + 1], t3[3 * i + 2]),
                               XYZ(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]), XYZ(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]));
        out3[3 * i] = r.X; out3[3 * i + 1] = r.Y; out3[3 * i + 2] = r.Z;
    }
}
void sro_apply_rotation(const float* p3, const float* q4, uint32_t n, float* out3) {
    for (uint32_t i = 0; i < n; i++) {
        Quat q; q.X = q4[4 * i]; q.Y = q4[4 * i + 1]; q.Z = q4[4 * i + 2]; q.W = q4[4 * i + 3];
        XYZ r = applyRotation(XYZ(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), q);
        out3[3 * i] = r.X; out3[3 * i + 1] = r.Y; out3[3 * i + 2] = r.Z;
    }
}

// output stage (sro_post.h): rgb3 = n linear pixels; out3 = clamp(..)*255 as floats, out_rgba8 = what the canvas stores
void sro_post_process(const float* rgb3, uint64_t n, int tonemap, float* out3, uint8_t* out_rgba8) {
    for (uint64_t i = 0; i < n; i++) {
        sro::RGB c = sro::post_process_pixel(sro::RGB{rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]}, tonemap);
        if (out3) { out3[3 * i] = c.X; out3[3 * i + 1] = c.Y; out3[3 * i + 2] = c.Z; }
        if (out_rgba8) {
            out_rgba8[4 * i] = sro::post_to_u8(c.X); out_rgba8[4 * i + 1] = sro::post_to_u8(c.Y);
            out_rgba8[4 * i + 2] = sro::post_to_u8(c.Z); out_rgba8[4 * i + 3] = 255;
        }
    }
}

}  // extern "C"
