/* include/srx.h — C-ABI of the MI355X trace-and-shade path ("srx").
 *
 * The reference (CharlesCowdery/RayTracing, directory spatialPartitioning/) has
 * no FFI: its seam is the header-only C++ class `RayEngine` (Core.h:36-48). This
 * C-ABI sits underneath that seam; every entry point below names the reference
 * interface it replaces. Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions: every function returns 0 on success, non-zero on failure, and
 * `srx_last_error()` then describes the failure (the reference throws C++
 * exceptions instead: Primitives.h:34, BVH.h:1040). Nothing is thrown across
 * this boundary. "device pointer" = HIP device memory on the current device.
 * A scene handle is immutable after creation and may be shared by threads; a
 * renderer handle owns scratch queues and must be used by one thread at a time.
 */
#ifndef SRX_H
#define SRX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct srx_package srx_package;
typedef struct srx_scene srx_scene;
typedef struct srx_renderer srx_renderer;

/* Authoring triangle in world space = one `Tri` (Primitives.h:224-258) after
 * `Object::fetchData` flattening (Primitives.h:423-434). flags bit0: vertex
 * normals `n` are given (else the face normal is used, as the Tri ctor does). */
typedef struct srx_tri {
    float p[9];
    float uv[6];
    float n[9];
    int32_t material;
    uint32_t flags;
} srx_tri;

/* One `Material` (Materials.h:317-358) as plain data: each parameter is a
 * constant or, when its *_tex id is >= 0, a texture (Texture.h:306-313). */
typedef struct srx_material {
    float color[3];     int32_t color_tex;
    float roughness;    int32_t roughness_tex;
    float metallic;     int32_t metallic_tex;
    float specular;     int32_t specular_tex;
    float emissive[3];  int32_t emissive_tex;
    float normal[3];    int32_t normal_tex;
    float transmission; int32_t transmission_tex;
    float ior;          int32_t ior_tex;
    int32_t use_normals;
} srx_material;

/* One `Texture<T>` (Texture.h:33-280, USE_CHAR 1): row-major RGB8 texels; a
 * float-valued texture keeps its value in byte 0 of each texel. */
typedef struct srx_texture {
    const uint8_t* rgb8;
    int32_t width, height;
    float scale_x, scale_y;       /* KHR_texture_transform scale (Texture.h:73,227-230) */
    int32_t normal_transform;     /* (c-128)/127 decode (Texture.h:151-153) */
} srx_texture;

/* `SunLight` (Primitives.h:39-51): `direction` is what vectorTo() returns (not normalised). */
typedef struct srx_sun {
    float direction[3];
    float emission[3];
} srx_sun;

/* `PackagedScene` (Scene.h:18-31) as pointers + counts, records in the reference
 * layouts: BVH_AVX 288 B (BVH.h:874-881), PTri_AVX 448 B (Primitives.h:127-134),
 * PackagedTri 196 B (Primitives.h:62-81). Everything is deep-copied at create. */
typedef struct srx_scene_desc {
    const void* nodes;     uint32_t n_nodes;
    const void* packets;   uint32_t n_packets;
    const void* tris;      uint32_t n_tris;
    const srx_material* materials; uint32_t n_materials;
    const srx_texture* textures;   uint32_t n_textures;
    const srx_sun* suns;           uint32_t n_suns;
    int32_t max_generations;       /* Scene.h:28 */
    int32_t monte_carlo_max;       /* Scene.h:29 */
    float monte_carlo_modifier;    /* Scene.h:30 */
    const float* pdf_lookup;       /* 1024*1024 floats (PDF.h:83-123) or NULL: built by the library */
    const float* arccos_lookup;    /* 2048 floats (VecLib.cpp:391-398) or NULL: built by the library */
} srx_scene_desc;

/* Bit-compatible subset of `CastResults` (Coredefs.h:6-15): miss = material 0xFFFFFFFF, t=u=v=-1, tri_index 0. */
typedef struct srx_hit {
    uint32_t material;
    float t, u, v;
    int32_t tri_index;
} srx_hit;

/* `Camera` + `RectLens` (Camera.h:32-140). */
typedef struct srx_camera {
    float position[3];
    float rotation[4];            /* quaternion x y z w */
    float focal_position[3];
    float lens_width, lens_height;
    int32_t res_x, res_y, subdiv; /* subdiv^2 strata per pixel */
} srx_camera;

/* Counters of one call; `rays_processed`/`diffuses_cast` are `Casting_Diagnostics` (Coredefs.h:17-23). */
typedef struct srx_counters {
    uint64_t rays_processed;      /* process_ray invocations (Core.h:360) */
    uint64_t diffuses_cast;       /* child rays queued (Core.h:595) */
    uint64_t shadows_cast;        /* execute_lighting_cast calls (Core.h:351) */
    uint64_t nodes;               /* 8-wide node tests by closest-hit casts (Core.h:70-72) */
    uint64_t packets;             /* 8-triangle packet tests by closest-hit casts (Core.h:194-229) */
    uint64_t shadow_nodes;
    uint64_t shadow_packets;
    uint64_t hits;                /* shaded hits */
    uint64_t texels;              /* texture fetches */
    uint64_t pdf_lookups;         /* NDF table fetches */
    uint64_t trace_launches;      /* closest-hit kernel launches in this call */
    double trace_ms;              /* summed closest-hit kernel time, HIP events on the launch stream (0 unless timing is on) */
    uint64_t shadow_launches;
    double shadow_ms;
    uint64_t shade_launches;
    double shade_ms;
} srx_counters;

const char* srx_last_error(void);

/* Select the HIP device for this thread and check that it is gfx950-class. */
int srx_init(int device);

/* ---- host packager: replaces `Scene::package()` (Scene.h:90-179) = BVH::construct (BVH.h:98-180)
 * + BVH_AVX::collapse (BVH.h:989-1077) + the PTri_AVX regroup (Scene.h:142-173). */
int srx_package_create(const srx_tri* tris, uint32_t n_tris, srx_package** out);
int srx_package_get(const srx_package* pkg, const void** nodes, uint32_t* n_nodes, const void** packets,
                    uint32_t* n_packets, const void** tris, uint32_t* n_tris);
void srx_package_destroy(srx_package* pkg);
/* A DIFFERENT tree over the same triangles, built fast: parallel binned SAH, tight leaf boxes, greedy
 * 8-wide collapse; same record formats, so srx_package_get / srx_scene_create take it as they take the other. Opt-in: the
 * reference's build (BVH.h:98-180, :581-653) defines the numbering of tri_data, and a different tree numbers the triangles
 * differently; it also leaves out the reference's origin-grown leaf boxes (quirk Q1) and the invisibility of exactly flat
 * axis-aligned subtrees (Q3). Closest hits are the same triangles with the same t, u, v except where two triangles tie. */
int srx_package_create_fast(const srx_tri* tris, uint32_t n_tris, srx_package** out);
/* order[k] = index of the input triangle that tri_data[k] (and so a hit's tri_index k) was packed from, for a package of
 * either builder: the remap between the two numberings. */
int srx_package_get_order(const srx_package* pkg, const uint32_t** order, uint32_t* n);

/* Lookup tables: `VecLib::prep` arccos table (VecLib.cpp:391-398) and `PDF::prep` (PDF.h:83-123). */
int srx_build_tables(float* arccos2048, float* pdf1m);

/* ---- scene upload: replaces `RayEngine::load_scene(PackagedScene*)` (Core.h:46-48). */
int srx_scene_create(const srx_scene_desc* desc, srx_scene** out);
void srx_scene_destroy(srx_scene* scene);
/* worst-case traversal stack depth of this scene's tree (the reference reserves 256, Scene.h:181) */
int srx_scene_stack_bound(const srx_scene* scene);
/* Times a traversal wave moved part of its stack from on-chip memory to the scene's overflow areas so far (deep trees
 * keep only the top of the stack on chip; no counterpart in the reference, whose stack is a 256-entry array). */
int64_t srx_scene_stack_spills(const srx_scene* scene);

/* ---- casts: replace `RayEngine::execute_ray_cast(XYZ&, const XYZ&)` (Core.h:346) over a batch.
 * Host-pointer form copies in and out; the _device form takes device pointers and a hipStream_t
 * (as void*) and is asynchronous. org3/dir3 are n*3 floats; `advanced_org3` (optional) receives the
 * origin after the reference's in-place advance `o += t*d*0.999f` (Core.h:327).
 * counters2 (optional, host) receives {node tests, packet tests}. */
int srx_cast_closest(srx_scene* scene, const float* org3, const float* dir3, uint32_t n, srx_hit* hits,
                     float* advanced_org3, uint64_t* counters2);
int srx_cast_closest_device(srx_scene* scene, const float* d_org3, const float* d_dir3, uint32_t n, srx_hit* d_hits,
                            uint64_t* d_counters2, void* stream);
/* srx_cast_closest_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc). NO COUNTERPART IN THE REFERENCE, whose casts use the CPU stack (Core.h:55-58).
 * Same kernels and results as srx_cast_closest_device; what that call keeps in memory allocated with the scene (the
 * traversal stack's overflow areas and their control words) and its counters (here replicated over 256 slots and folded
 * into d_counters2 afterwards, so counting does not serialise on one word) live in a static device array of the
 * library: 64.1 MB per GPU the library runs on, held until the process ends. Limit: scenes whose traversal stack bound
 * (srx_scene_stack_bound) exceeds 64 entries are refused. Allocates nothing, frees nothing; asynchronous on `stream`.
 * All calls on one GPU share that array: calls that overlap on one GPU (two streams, two threads) race. d_counters2
 * (optional, device, zeroed by the caller) is added to, as in srx_cast_closest_device. */
int srx_cast_closest_device_static(srx_scene* scene, const float* d_org3, const float* d_dir3, uint32_t n, srx_hit* d_hits,
                                   uint64_t* d_counters2, void* stream);
/* Diagnostic of the call above (no counterpart in the reference either): waits for `stream` and reports in *truncated
 * whether a traversal wave of the last static call on the current GPU found no free overflow area and ended its rays
 * early. That cannot happen with one call at a time; overlapping calls on one GPU (which the call above says race) can
 * cause it, and a caller that cannot rule them out must check this before trusting d_hits. Allocates nothing. */
int srx_static_scratch_status(void* stream, int32_t* truncated);
/* replace `RayEngine::execute_lighting_cast(XYZ, Light*)` (Core.h:351-358) for sun lights:
 * lit[i] = 1 iff the closest hit along dir is farther than 9999 (Primitives.h:44-50). */
int srx_cast_shadow(srx_scene* scene, const float* org3, const float* dir3, uint32_t n, uint8_t* lit);
int srx_cast_shadow_device(srx_scene* scene, const float* d_org3, const float* d_dir3, uint32_t n, uint8_t* d_lit,
                           void* stream);

/* ---- wavefront renderer: replaces the `RenderThread::process_iteratively` loop around
 * `RayEngine::process_ray` (SceneManager.h:193-237, Core.h:359-601).
 * `ray_capacity` = number of ray slots in the device queues (48 B ray + 20 B hit + shadow records each). */
int srx_renderer_create(srx_scene* scene, uint64_t ray_capacity, srx_renderer** out);
void srx_renderer_destroy(srx_renderer* r);
/* 1 = bracket kernels with HIP events on the launch stream and report ms in srx_counters */
int srx_renderer_set_timing(srx_renderer* r, int enabled);

/* One sample pass over the pixel rectangle [x0,x0+w) x [y0,y0+h): primary rays from `camera`
 * (stratum pass_index % subdiv^2, jitter from the per-sample keyed stream), full branched shading,
 * then the running-mean accumulation of SceneManager.h:224-229 into d_accum4 and (odd passes) d_aux4:
 * device arrays of res_x*res_y float4 = (r, g, b, sample count). Non-finite samples are skipped.
 * Synchronous: returns when the pass is complete. `counters` optional (host). */
int srx_trace_pass(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                   uint32_t pass_index, uint64_t seed, float* d_accum4, float* d_aux4, srx_counters* counters);

/* The same pass over this caller's share of the rectangle's rows: groups of `group_rows` rows dealt round-robin
 * over `nparts` callers (one per GPU), this caller taking groups part, part + nparts, ...; the pixels of the other
 * callers are not touched. The reference deals interleaved pixel blocks to its render threads the same way
 * (SceneManager::assign_iterative_groups, SceneManager.h:441-459), which evens out cheap and expensive image
 * regions. srx_trace_pass == group_rows h, part 0, nparts 1. Results per pixel do not depend on the split. */
int srx_trace_pass_interleaved(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                               int32_t group_rows, int32_t part, int32_t nparts, uint32_t pass_index, uint64_t seed,
                               float* d_accum4, float* d_aux4, srx_counters* counters);
/* n_passes consecutive sample passes (pass_index .. pass_index + n_passes - 1) of the same rectangle and share in ONE
 * call: the rays of all of them travel through the same kernel launches (what the reference gets from keeping every
 * core busy on its own block, SceneManager.h:137-165, a GPU gets from launches that are large enough: a 1/8 share of
 * a 1080p frame runs 12 % below the full frame's rate with one pass per call). Same result as n_passes calls of
 * srx_trace_pass_interleaved, sample for sample; pixels x n_passes < 2^24. */
int srx_trace_passes_interleaved(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                                 int32_t group_rows, int32_t part, int32_t nparts, uint32_t pass_index, int32_t n_passes,
                                 uint64_t seed, float* d_accum4, float* d_aux4, srx_counters* counters);

/* ---- adaptive sampling: replaces SceneManager::analyze_noise (SceneManager.h:526-667), the
 * priority half of SceneManager::balance_threads (:668-692) and the per-block visit rule of
 * RenderThread::process_iteratively (:193-198). Blocks are the reference's 4x4 pixel blocks (:299), numbered
 * row-major over ceil(res_x/4) x ceil(res_y/4). */
typedef struct srx_noise srx_noise;
typedef struct srx_noise_region {          /* SceneManager::noise_region, SceneManager.h:262-273 */
    int32_t x0, y0, x1, y1;                /* start (inclusive), end (exclusive) */
    int32_t active;
    float noise, max_noise, min_noise, avg_noise;
} srx_noise_region;

/* one region covering the frame (SceneManager.h:937-938); minimum_samples < 0 = the reference's 64 (:301) */
int srx_noise_create(int32_t res_x, int32_t res_y, uint64_t seed, int32_t minimum_samples, srx_noise** out);
void srx_noise_destroy(srx_noise* c);
/* Options that leave the reference's behaviour (all off by default = the parity path).
 * SRX_NOISE_ABS_LUMINANCE: the error term |L - L_aux| / max(L^1.1, L) (:568-572) takes |L| in the denominator. In the
 * reference a pixel with negative luminance (direct light on a surface that faces away from the sun is negative,
 * Core.h:461-463) makes pow(L, 1.1f) NaN, then its region's noise, then every block priority: sampling stops on any
 * sunlit scene (quirk Q11). With the option on such pixels count like positive ones. */
#define SRX_NOISE_ABS_LUMINANCE 1
int srx_noise_set_option(srx_noise* c, int32_t option, int32_t value);
/* device: d_err[pixel] = |L - L_aux| / max(L^1.1, L) (:568-572; 0 where L == 0) for the rows this caller owns
 * (interleave as in srx_trace_pass_interleaved) and 0 elsewhere, so a sum over the callers completes the map;
 * *min_samples = the smallest per-pixel sample count among the owned rows (:492-501). */
int srx_noise_error_map(srx_noise* c, const float* d_accum4, const float* d_aux4, int32_t group_rows, int32_t part,
                        int32_t nparts, float* d_err, int32_t* min_samples);
/* analyze_noise on the complete error map (device pointer): region statistics on the device, split search and
 * region list on the host. `seen` = minimum sample count over the whole frame. */
int srx_noise_analyze(srx_noise* c, const float* d_err, int32_t seen);
/* copies at most `capacity` regions, *count = how many exist */
int srx_noise_regions(const srx_noise* c, srx_noise_region* out, uint32_t capacity, uint32_t* count);
/* block priorities (:669-692) into d_block_priority (device, one float per block); all 1 while seen < minimum_samples
 * (*updated = 0), as the reference leaves block::priority at its initial value (:36). */
int srx_noise_priorities(srx_noise* c, int32_t seen, float* d_block_priority, int32_t* updated);

/* One VISIT of every 4x4 block of this caller's rows (RenderThread::process_iteratively, SceneManager.h:193-237):
 * a block takes floor(p) + [u < frac(p)] samples (0 or 1; p <= 1 by construction) with u keyed by (seed, visit_index,
 * block); a pixel's stratum and running-mean weight come from its own count so far (block->iterations_done), its
 * random stream from (seed, visit_index, pixel), and the aux buffer takes the odd counts (:226-233). d_block_priority NULL = all 1. With all priorities 1 and equal counts this is srx_trace_pass. */
int srx_trace_pass_adaptive(srx_renderer* r, const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h,
                            int32_t group_rows, int32_t part, int32_t nparts, uint32_t visit_index, uint64_t seed,
                            const float* d_block_priority, float* d_accum4, float* d_aux4, srx_counters* counters);

/* `process_ray` on an explicit batch of rays (host arrays): ray i starts at generation `generation`
 * with keyed RNG state keys[i]; radiance3[i] receives the returned XYZ. Used by parity tests. */
int srx_process_rays(srx_renderer* r, const float* org3, const float* dir3, const uint64_t* keys, int32_t generation,
                     uint32_t n, float* radiance3, srx_counters* counters);

/* Primary ray directions + post-jitter RNG keys for a pixel rectangle (host outputs):
 * `Camera::random_slope_at` (Camera.h:138-140) on the device. */
int srx_camera_rays(const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h, uint32_t pass_index,
                    uint64_t seed, float* dir3, uint64_t* keys);

/* The same with the REFERENCE's generator (`XorGen`, XorRandom.cpp:3-69, 119-132; `thread_local XorGen gen`, commons.cpp:30):
 * ONE sequential stream seeded with seed4 (XorGen::seed), two `fRand` draws per pixel (Camera.h:95), pixels in row-major
 * order of the rectangle: what one reference thread draws when it walks the rectangle. These are the parity ray sets of
 * fixed XorRandom seeds; the renderer itself draws from per-ray keyed streams. At most 2^22 rays. */
int srx_camera_rays_xorgen(const srx_camera* camera, int32_t x0, int32_t y0, int32_t w, int32_t h, uint32_t pass_index,
                           const uint64_t* seed4, float* dir3);
/* Raw streams of that generator, computed on the device: n_pools x `regen_pool` (8 floats each, XorRandom.cpp:35-69), then
 * n_scalar x `fRand(0, 1)` (:128-132), from one XorGen seeded with seed4. For parity tests of the generator. */
int srx_xorgen_stream(const uint64_t* seed4, uint32_t n_pools, uint32_t n_scalar, float* pools8, float* scalars);

/* ---- device buffers for a host without a HIP toolchain. The reference keeps its framebuffers on the host
 * (SceneManager::raw_output / aux_output, SceneManager.h:288-289, filled at :370-380); here they live on the device as
 * res_x * res_y float4 {running mean r, g, b, sample count}. srx_buffer_create returns zeroed device memory (a
 * framebuffer is res_x * res_y * 16 bytes, a canvas for srx_post_process res_x * res_y * 4), srx_buffer_read copies it
 * to the host (srx_buffer_write the other way), srx_buffer_clear zeroes it again (`SceneManager::reset_blocks`, SceneManager.h:478-487). */
int srx_buffer_create(uint64_t bytes, void** d_out);
void srx_buffer_destroy(void* d_buf);
int srx_buffer_read(const void* d_buf, uint64_t bytes, void* host_out);
int srx_buffer_write(void* d_buf, uint64_t bytes, const void* host_in);
int srx_buffer_clear(void* d_buf, uint64_t bytes);

/* ---- output stage ---------------------------------------------------------------------------
 * srx_post_process replaces the canvas conversion of SceneManager::refresh_canvas (SceneManager.h:1064-1072):
 * ImageHandler::post_process_pixel (ImageProcessing.h:137-151: luminance-preserving gamma 1, view transform,
 * luminance-preserving gamma 2.2, HSV round trip, clamp to [0, 1] x 255) on every pixel of the float4 framebuffer,
 * then GUI::commit_pixel's float -> 8-bit store (GUI.h:104-107, truncation). d_rgba8 (4 B per pixel, alpha 255) and
 * d_rgb_out (3 floats per pixel in [0, 255]) are device pointers; either may be NULL.
 * The reference's view transform is an OpenColorIO processor built from a config file that is not in its repository
 * (:9-11): SRX_TONEMAP_NONE leaves that step out (the parity path), SRX_TONEMAP_AGX_APPROX puts the public analytic
 * AgX approximation in its place. */
#define SRX_TONEMAP_NONE 0
#define SRX_TONEMAP_AGX_APPROX 1
int srx_post_process(const float* d_accum4, int32_t res_x, int32_t res_y, int32_t tonemap, uint8_t* d_rgba8,
                     float* d_rgb_out);
/* FileManager::writeRawFile / writeRaw (FileLoading.h:166-182): u32 res_x, u32 res_y, then res_y rows of res_x float3
 * pixels (row 0 first) from the RGB of the float4 framebuffer (device pointer). */
int srx_write_raw(const char* path, const float* d_accum4, int32_t res_x, int32_t res_y);
/* FileManager::readRaw (FileLoading.h:183-199) into a host array of 3 floats per pixel; rgb NULL = header only. */
int srx_read_raw(const char* path, float* rgb, uint64_t capacity_pixels, int32_t* res_x, int32_t* res_y);
/* The canvas as an 8-bit RGBA PNG (the reference shows it in an SFML window, GUI.h:104-125; there is no file
 * writer to mirror): stored deflate blocks, no compression library. d_rgba8 = device pointer from srx_post_process. */
int srx_write_png(const char* path, const uint8_t* d_rgba8, int32_t res_x, int32_t res_y);

/* ---- scene ingestion: replaces FileManager::loadGLTFFile / parseGLTFData (FileLoading.h:200-224,
 * 438-782) and the flattening the reference applies before Scene::package (Object::fetchData, Primitives.h:423-434):
 * a binary glTF (.glb) becomes world-space srx_tri records (in the order Scene::package would collect them), the
 * material table (the file's materials + the reference's "SR default material" last), textures as the reference
 * stores them (Texture.h, 3 bytes per texel; embedded 8-bit PNG and baseline JPEG images), sun lights and the first camera
 * (res_x, res_y, subdiv are left 0 for the caller). Axis conventions, factors and quirks: see csrc/srx_gltf.cpp.
 * Host-only; the arrays belong to the handle. */
typedef struct srx_gltf srx_gltf;
int srx_gltf_load(const char* path, srx_gltf** out);
void srx_gltf_destroy(srx_gltf* g);
int srx_gltf_get(const srx_gltf* g, const srx_tri** tris, uint32_t* n_tris, const srx_material** materials, uint32_t* n_materials,
                 const srx_texture** textures, uint32_t* n_textures, const srx_sun** suns, uint32_t* n_suns, srx_camera* camera);
/* The image decoder the loader uses, on its own: replaces stbi_load_from_memory(..., req_comp = 4) as tinygltf's
 * LoadImageData calls it for every embedded image (FileLoading.h:438 goes through tinygltf::TinyGLTF::LoadBinaryFromFile;
 * stb_image.h:1399, 3865-4026 for JPEG). PNG of every colour type and bit depth (1-16), plain or Adam7-interlaced, with
 * tRNS; baseline, extended-sequential and progressive Huffman JPEG (1 or 3 components, any sampling factors that divide
 * the largest); the bytes equal stb_image's. Arithmetic-coded, lossless, 12-bit and CMYK JPEG are refused. Writes width and
 * height; when rgba is not NULL also width * height * 4 bytes (refused if capacity is smaller). Host-only. */
int srx_image_decode(const uint8_t* data, size_t n, uint32_t* width, uint32_t* height, uint8_t* rgba, size_t capacity);

/* ---- diagnostics (host only; no counterpart in the reference)
 * How srx_cast_* / the renderer deal n rays to the workgroups of the traversal kernel form with k (8, 4, 2, 1) rays per
 * octet: full blocks of 8 k rays, then up to tail_blocks blocks each of 4 k, 2 k, ... 8 rays, so that a launch does not
 * end with a few long waves (csrc/srx_kernels.h). Fills start[] / count[] for up to max_blocks blocks; *n_blocks = the
 * blocks used, *bound = the grid taken for a launch that may hold up to n rays. Tests check the ranges tile [0, n). */
int srx_trace_dealing(uint32_t n, int32_t k, uint32_t tail_blocks, uint32_t max_blocks, uint32_t* start, uint32_t* count,
                      uint32_t* n_blocks, uint32_t* bound);
/* ... and which logical block each hardware block of a launch of nblk (a multiple of 8) blocks takes: XCD b % 8 works through
 * interleaved chunks of `chunk` logical blocks. logical[nblk]; a permutation of 0 .. nblk - 1. */
int srx_trace_xcd_map(uint32_t nblk, uint32_t chunk, uint32_t* logical);

/* ---- noise-weighted dealing of the frame's row groups to the ranks: replaces the second half of
 * SceneManager::balance_threads (SceneManager.h:693-773), which after every analysis moves 4x4 blocks from the most
 * burdened render thread's list to the least burdened one's until their burdens (sums of block priorities) are within
 * 5 %. Here the unit is a group of group_rows pixel rows (a multiple of 4) and a part is a rank. A deal starts as the
 * round-robin of srx_trace_pass_interleaved (assign_iterative_groups, :441-459). srx_deal_rebalance is a function of the
 * block priorities alone (HOST array as srx_noise_priorities wrote them, copied back), which every rank holds
 * identically after the error-map exchange: every rank arrives at the same deal, no collective. A rank that takes over
 * rows needs their accumulated pixels: gather the framebuffers under the OLD deal before rebalancing. Host functions;
 * no GPU needed for create / rebalance / get. */
typedef struct srx_deal srx_deal;
int srx_deal_create(int32_t res_x, int32_t res_y, int32_t group_rows, int32_t nparts, srx_deal** out);
void srx_deal_destroy(srx_deal* d);
int srx_deal_rebalance(srx_deal* d, const float* block_priority, uint32_t n_blocks, int32_t* groups_moved);
/* owner[g] = the part group g belongs to, slot[g] = its index in that part's list (a part renders, packs and unpacks its
 * groups in list order); *n_groups = ceil(res_y / group_rows); burdens[nparts] = each part's burden at the last rebalance. */
int srx_deal_get(const srx_deal* d, int32_t* owner, int32_t* slot, uint32_t capacity, uint32_t* n_groups, float* burdens);
/* the dealt forms of srx_trace_pass_adaptive (whole frame), srx_noise_error_map, the gathers and the row packing */
int srx_trace_pass_adaptive_dealt(srx_renderer* r, const srx_camera* camera, srx_deal* deal, int32_t part, uint32_t visit_index, uint64_t seed,
                                  const float* d_block_priority, float* d_accum4, float* d_aux4, srx_counters* counters);
int srx_noise_error_map_dealt(srx_noise* c, const float* d_accum4, const float* d_aux4, srx_deal* deal, int32_t part, float* d_err,
                              int32_t* min_samples);
uint64_t srx_deal_slot_floats(const srx_deal* d, int32_t floats_per_pixel);
int srx_rows_pack_dealt(srx_deal* d, const float* d_frame, int32_t floats_per_pixel, int32_t part, float* d_slot, void* stream);
int srx_rows_unpack_dealt(srx_deal* d, const float* d_slots, uint64_t slot_stride_floats, uint64_t buffer_offset_floats, int32_t floats_per_pixel,
                          int32_t skip_part, float* d_frame, void* stream);

/* The shading path's device math over arrays, for parity tests against the reference's own object code (no render path
 * calls it): per element i, with a = a3[i], b = b3[i], c = c3[i], q = q4[i], out26[26 i ..] =
 *   XYZ::reflect(a, b) [0..2] (XYZ.cpp:178-186), cross(a, b) [3..5] (:292), normalize(a) [6..8] (:237), dot(a, b) [9] (:279),
 *   cdot(a, b) [10] (:282), Quat::applyRotation(a, q) [11..13] (:679-683), linear_mix(s, a, b) [14..16] (:397),
 *   Matrix3x3::applyTBNMatrix(a, T = b, B = c, N = q.xyz) [17..19] (Matrix.h:62-64), VecLib::full_sin / full_cos / tan of a.x
 *   [20..22] (VecLib.cpp:186-267), `s - a` with quirk Q4 [23..25] (XYZ.cpp:431-433). Host pointers. */
int srx_math_probe(const float* a3, const float* b3, const float* c3, const float* q4, float s, uint32_t n, float* out26);

/* ---- multi-GPU exchange: one process per GPU, scene replicated, tile rows dealt round-robin
 * (srx_trace_pass_interleaved), nothing exchanged on the sample path. What the reference's worker threads hand over in
 * shared memory becomes a collective over RCCL / xGMI (RCCL is loaded at run time; single-GPU use does not need it):
 *   SceneManager::get_screen_update (SceneManager.h:801-815: every block's pixels into one output)  -> srx_group_gather_framebuffer
 *   the per-pixel error terms analyze_noise sums (:559-579)                                         -> srx_group_allgather_rows on the error map
 *   SceneManager::gather_thread_stats (:488-502: minimum samples) and counters                      -> srx_group_allreduce
 * Bootstrap: rank 0 calls srx_group_unique_id and hands the 128 bytes to the other ranks by any means (a file, MPI,
 * torch.distributed); every rank then calls srx_group_create with its GPU current. Calls are collective (every rank
 * makes them in the same order) and return when the exchange is complete. */
typedef struct srx_group srx_group;
int srx_group_unique_id(void* id128);
int srx_group_create(const void* id128, int32_t rank, int32_t nranks, srx_group** out);
void srx_group_destroy(srx_group* g);
int srx_group_rank(const srx_group* g, int32_t* rank, int32_t* nranks);
/* In-place all-gather of float4 framebuffers (device) whose rows of THIS rank are valid, rows dealt in groups of
 * group_rows (rank = part, nranks = nparts of srx_trace_pass_interleaved): main and aux buffer (d_aux4 may be NULL) travel
 * in ONE collective, each rank's rows once. */
int srx_group_gather_framebuffer(srx_group* g, float* d_accum4, float* d_aux4, int32_t res_x, int32_t res_y, int32_t group_rows);
/* The same for any per-pixel device array of floats_per_pixel floats (the error map of srx_noise_error_map: 1). After it
 * every rank holds the complete map and srx_noise_analyze gives the same regions on every rank as on one GPU. */
int srx_group_allgather_rows(srx_group* g, float* d_frame, int32_t floats_per_pixel, int32_t res_x, int32_t res_y, int32_t group_rows);
/* In-place all-reduce over the ranks: d_sums[n_sums] doubles are summed, d_mins[n_mins] int32 take the minimum (device). */
/* the two gathers under a deal (nparts of the deal = ranks of the group, part = rank) */
int srx_group_gather_framebuffer_dealt(srx_group* g, srx_deal* deal, float* d_accum4, float* d_aux4);
int srx_group_allgather_rows_dealt(srx_group* g, srx_deal* deal, float* d_frame, int32_t floats_per_pixel);
int srx_group_allreduce(srx_group* g, double* d_sums, uint32_t n_sums, int32_t* d_mins, uint32_t n_mins);
/* The row packing on its own, for a host that brings its own transport: a part's rows of a frame packed into a slot of
 * srx_rows_slot_floats() floats per frame (the size of the largest share, so slots are equal), and all parts' slots
 * (slot_stride_floats apart, this frame at buffer_offset_floats inside each) scattered back into a frame; skip_part's rows
 * are left as they are (-1: none). `stream` = hipStream_t, asynchronous. */
uint64_t srx_rows_slot_floats(int32_t floats_per_pixel, int32_t res_x, int32_t res_y, int32_t group_rows, int32_t nparts);
int srx_rows_pack(const float* d_frame, int32_t floats_per_pixel, int32_t res_x, int32_t res_y, int32_t group_rows, int32_t part,
                  int32_t nparts, float* d_slot, void* stream);
int srx_rows_unpack(const float* d_slots, uint64_t slot_stride_floats, uint64_t buffer_offset_floats, int32_t floats_per_pixel, int32_t res_x,
                    int32_t res_y, int32_t group_rows, int32_t nparts, int32_t skip_part, float* d_frame, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
