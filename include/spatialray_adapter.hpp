// include/spatialray_adapter.hpp — the reference's seam, re-expressed over the srx C-ABI (header only, C++17).
//
// A render loop written against the reference calls (Core.h:36-48, :346-359):
//     RayEngine RE;  RE.load_scene(scene->package());
//     CastResults r = RE.execute_ray_cast(position, slope);
//     bool lit     = RE.execute_lighting_cast(position, light);
//     XYZ radiance = RE.process_ray(stats, ray);
// This header keeps those names and argument meanings. Single-ray calls go through the batched entry points
// with n = 1 (correct, slow: a GPU wants batches); the *_batch members are what a ported loop should call.
// Errors: the reference throws C++ exceptions; the C-ABI returns codes, the adapter rethrows std::runtime_error.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "srx.h"

namespace spatialray {

struct XYZ { float X = 0, Y = 0, Z = 0; };                       // XYZ.h:49-53
struct XY { float X = 0, Y = 0; };

struct CastResults {                                             // Coredefs.h:6-15
    uint32_t material = 0xFFFFFFFFu;
    float distance = 99999999999999999.f;
    XYZ TUV{-1, -1, -1};
    XY texUV{-1, -1};
    int tri_index = 0;
};
struct Casting_Diagnostics { long long rays_processed = 0; long diffuses_cast = 0; long shadows_cast = 0; };   // Coredefs.h:17-23
struct PackagedRay { XYZ position, slope; char generation = 0; XYZ coefficient{1, 1, 1}; uint64_t key = 0; };  // Ray.h:7-25
struct SunLight { XYZ rotation, emission; };                     // Primitives.h:39-51

inline void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + srx_last_error());
}

// Scene::package() (Scene.h:90-179): owns the three flat arrays in the reference's layouts.
class PackagedScene {
public:
    std::vector<srx_material> materials;
    std::vector<srx_texture> textures;
    std::vector<srx_sun> lights;
    short max_generations = 10, monte_carlo_max = 32;            // Scene.h:27-30
    float monte_carlo_modifier = 1.0f / 4;
    explicit PackagedScene(const std::vector<srx_tri>& tris) {
        check(srx_package_create(tris.data(), (uint32_t)tris.size(), &pkg_), "srx_package_create");
    }
    ~PackagedScene() { srx_package_destroy(pkg_); }
    PackagedScene(const PackagedScene&) = delete;
    PackagedScene& operator=(const PackagedScene&) = delete;
    srx_scene_desc desc() const {
        srx_scene_desc d;
        std::memset(&d, 0, sizeof(d));
        check(srx_package_get(pkg_, &d.nodes, &d.n_nodes, &d.packets, &d.n_packets, &d.tris, &d.n_tris), "srx_package_get");
        d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
        d.textures = textures.data(); d.n_textures = (uint32_t)textures.size();
        d.suns = lights.data(); d.n_suns = (uint32_t)lights.size();
        d.max_generations = max_generations; d.monte_carlo_max = monte_carlo_max; d.monte_carlo_modifier = monte_carlo_modifier;
        return d;
    }
private:
    srx_package* pkg_ = nullptr;
};

class RayEngine {                                                // Core.h:36-48
public:
    explicit RayEngine(int device = 0, uint64_t ray_capacity = 1u << 22) : capacity_(ray_capacity) { check(srx_init(device), "srx_init"); }
    ~RayEngine() { srx_renderer_destroy(renderer_); srx_scene_destroy(scene_); }
    RayEngine(const RayEngine&) = delete;
    RayEngine& operator=(const RayEngine&) = delete;

    void load_scene(const PackagedScene* PS) {                   // Core.h:46-48
        srx_renderer_destroy(renderer_); renderer_ = nullptr;
        srx_scene_destroy(scene_); scene_ = nullptr;
        const srx_scene_desc d = PS->desc();
        check(srx_scene_create(&d, &scene_), "srx_scene_create");
    }
    // Core.h:346 — `position` is advanced in place like the reference (o += t*d*0.999f, even on a miss).
    CastResults execute_ray_cast(XYZ& position, const XYZ& slope) {
        srx_hit h;
        float adv[3];
        check(srx_cast_closest(scene_, &position.X, &slope.X, 1, &h, adv, nullptr), "srx_cast_closest");
        position = XYZ{adv[0], adv[1], adv[2]};
        return to_results(h);
    }
    void execute_ray_cast_batch(const float* org3, const float* dir3, uint32_t n, srx_hit* hits, float* advanced = nullptr) {
        check(srx_cast_closest(scene_, org3, dir3, n, hits, advanced, nullptr), "srx_cast_closest");
    }
    bool execute_lighting_cast(XYZ position, const SunLight& L) {   // Core.h:351-358
        uint8_t lit = 0;
        check(srx_cast_shadow(scene_, &position.X, &L.rotation.X, 1, &lit), "srx_cast_shadow");
        return lit != 0;
    }
    // Core.h:359-601 — ray_data.key selects the ray's random stream (the reference uses a thread-local generator).
    XYZ process_ray(Casting_Diagnostics& stats, PackagedRay& ray_data) {
        float rad[3];
        srx_counters c;
        check(srx_process_rays(renderer(), &ray_data.position.X, &ray_data.slope.X, &ray_data.key, ray_data.generation, 1, rad, &c),
              "srx_process_rays");
        stats.rays_processed += (long long)c.rays_processed;
        stats.diffuses_cast += (long)c.diffuses_cast;
        stats.shadows_cast += (long)c.shadows_cast;
        return XYZ{rad[0] * ray_data.coefficient.X, rad[1] * ray_data.coefficient.Y, rad[2] * ray_data.coefficient.Z};
    }
    // RenderThread::process_iteratively (SceneManager.h:193-237) for a rectangle: device framebuffers.
    srx_counters trace_pass(const srx_camera& cam, int x0, int y0, int w, int h, uint32_t pass, uint64_t seed, float* d_accum4, float* d_aux4) {
        srx_counters c;
        check(srx_trace_pass(renderer(), &cam, x0, y0, w, h, pass, seed, d_accum4, d_aux4, &c), "srx_trace_pass");
        return c;
    }
    // one GPU's share of the pass: row groups part, part + nparts, ... (SceneManager::assign_iterative_groups, SceneManager.h:441-459)
    srx_counters trace_pass_interleaved(const srx_camera& cam, int x0, int y0, int w, int h, int group_rows, int part, int nparts,
                                        uint32_t pass, uint64_t seed, float* d_accum4, float* d_aux4) {
        srx_counters c{};
        check(srx_trace_pass_interleaved(renderer(), &cam, x0, y0, w, h, group_rows, part, nparts, pass, seed, d_accum4, d_aux4, &c),
              "srx_trace_pass_interleaved");
        return c;
    }
    // several consecutive passes in one call: their rays share the kernel launches (a share of a frame keeps the GPU full)
    srx_counters trace_passes_interleaved(const srx_camera& cam, int x0, int y0, int w, int h, int group_rows, int part, int nparts,
                                          uint32_t pass, int n_passes, uint64_t seed, float* d_accum4, float* d_aux4) {
        srx_counters c{};
        check(srx_trace_passes_interleaved(renderer(), &cam, x0, y0, w, h, group_rows, part, nparts, pass, n_passes, seed, d_accum4,
                                           d_aux4, &c), "srx_trace_passes_interleaved");
        return c;
    }
    srx_scene* scene() const { return scene_; }

private:
    srx_renderer* renderer() {
        if (!renderer_) check(srx_renderer_create(scene_, capacity_, &renderer_), "srx_renderer_create");
        return renderer_;
    }
    static CastResults to_results(const srx_hit& h) {
        CastResults r;
        r.material = h.material;
        if (h.material != 0xFFFFFFFFu) { r.distance = h.t; r.TUV = XYZ{h.t, h.u, h.v}; }
        r.tri_index = h.tri_index;
        return r;
    }
    srx_scene* scene_ = nullptr;
    srx_renderer* renderer_ = nullptr;
    uint64_t capacity_;
};

// The frame a render loop accumulates into: SceneManager::raw_output / aux_output (SceneManager.h:288-289, 370-380) as
// device float4 framebuffers {running mean r, g, b, sample count}; no HIP headers needed on the host side.
class FrameBuffers {
public:
    int res_x, res_y;
    float* raw_output = nullptr;     // device pointers
    float* aux_output = nullptr;
    FrameBuffers(int rx, int ry) : res_x(rx), res_y(ry) {
        check(srx_buffer_create(bytes(), (void**)&raw_output), "srx_buffer_create");
        check(srx_buffer_create(bytes(), (void**)&aux_output), "srx_buffer_create");
    }
    ~FrameBuffers() { srx_buffer_destroy(raw_output); srx_buffer_destroy(aux_output); }
    FrameBuffers(const FrameBuffers&) = delete;
    FrameBuffers& operator=(const FrameBuffers&) = delete;
    uint64_t bytes() const { return (uint64_t)res_x * (uint64_t)res_y * 16u; }
    void reset() {                                               // SceneManager::reset_blocks, SceneManager.h:478-487
        check(srx_buffer_clear(raw_output, bytes()), "srx_buffer_clear");
        check(srx_buffer_clear(aux_output, bytes()), "srx_buffer_clear");
    }
    std::vector<float> read(bool aux = false) const {            // res_x * res_y * {r, g, b, samples}
        std::vector<float> host((size_t)res_x * res_y * 4);
        check(srx_buffer_read(aux ? aux_output : raw_output, bytes(), host.data()), "srx_buffer_read");
        return host;
    }
};

// One rank of a multi-GPU render (one process per GPU): what SceneManager's threads hand over in shared memory
// (get_screen_update SceneManager.h:801-815, gather_thread_stats :488-502, the error terms of analyze_noise :559-579)
// travels through srx_group (RCCL). Rank 0 makes the id, the host passes its 128 bytes to the other ranks.
class Group {
public:
    static std::vector<char> unique_id() { std::vector<char> id(128); check(srx_group_unique_id(id.data()), "srx_group_unique_id"); return id; }
    Group(const std::vector<char>& id128, int rank, int nranks) {
        if (id128.size() != 128) throw std::runtime_error("Group: the id is 128 bytes");
        check(srx_group_create(id128.data(), rank, nranks, &g_), "srx_group_create");
        rank_ = rank; nranks_ = nranks;
    }
    ~Group() { srx_group_destroy(g_); }
    Group(const Group&) = delete;
    Group& operator=(const Group&) = delete;
    int rank() const { return rank_; }
    int nranks() const { return nranks_; }
    // every rank's tile rows (groups of group_rows rows dealt round-robin) into every rank's frame, main + aux in one collective
    void gather_framebuffer(float* d_accum4, float* d_aux4, int res_x, int res_y, int group_rows) {
        check(srx_group_gather_framebuffer(g_, d_accum4, d_aux4, res_x, res_y, group_rows), "srx_group_gather_framebuffer");
    }
    void allgather_rows(float* d_frame, int floats_per_pixel, int res_x, int res_y, int group_rows) {
        check(srx_group_allgather_rows(g_, d_frame, floats_per_pixel, res_x, res_y, group_rows), "srx_group_allgather_rows");
    }
    void allreduce(double* d_sums, uint32_t n_sums, int32_t* d_mins, uint32_t n_mins) {
        check(srx_group_allreduce(g_, d_sums, n_sums, d_mins, n_mins), "srx_group_allreduce");
    }
private:
    srx_group* g_ = nullptr;
    int rank_ = 0, nranks_ = 1;
};

// FileManager (FileLoading.h): scene in, raw frame out
namespace FileManager {
struct GltfScene {                                               // what parseGLTFData leaves in Scene, flattened (FileLoading.h:438-782)
    std::vector<srx_tri> tris; std::vector<srx_material> materials; std::vector<srx_texture> textures; std::vector<srx_sun> suns;
    srx_camera camera{};
    srx_gltf* handle = nullptr;                                  // owns the texture bytes the srx_texture records point to
    GltfScene() = default;
    GltfScene(const GltfScene&) = delete;
    GltfScene& operator=(const GltfScene&) = delete;
    ~GltfScene() { srx_gltf_destroy(handle); }
};
inline void loadGLTFFile(const std::string& fName, GltfScene& out) {   // FileLoading.h:200-224
    check(srx_gltf_load(fName.c_str(), &out.handle), "srx_gltf_load");
    const srx_tri* t; const srx_material* m; const srx_texture* x; const srx_sun* s; uint32_t nt, nm, nx, ns;
    check(srx_gltf_get(out.handle, &t, &nt, &m, &nm, &x, &nx, &s, &ns, &out.camera), "srx_gltf_get");
    out.tris.assign(t, t + nt); out.materials.assign(m, m + nm); out.textures.assign(x, x + nx); out.suns.assign(s, s + ns);
}
inline void writeRawFile(const float* d_accum4, int res_x, int res_y, const std::string& fName) {   // FileLoading.h:166-182
    check(srx_write_raw(fName.c_str(), d_accum4, res_x, res_y), "srx_write_raw");
}
}  // namespace FileManager

// ImageHandler::postProcessRaw (ImageProcessing.h:152-158) + the canvas store of GUI::commit_pixel (GUI.h:104-107)
namespace ImageHandler {
inline void postProcessRaw(const float* d_accum4, int res_x, int res_y, uint8_t* d_rgba8, int tonemap = SRX_TONEMAP_NONE) {
    check(srx_post_process(d_accum4, res_x, res_y, tonemap, d_rgba8, nullptr), "srx_post_process");
}
}  // namespace ImageHandler

}  // namespace spatialray
