// examples/render_glb.cpp — a headless render loop written against the reference's names, over the C-ABI.
//
// What the reference's Main.cpp:49-205 + SceneManager::render do for one frame, without the window: load a .glb
// (FileManager::loadGLTFFile, FileLoading.h:200-224), package it (Scene::package, Scene.h:90-179), hand it to the engine
// (RayEngine::load_scene, Core.h:46-48), take `spp` sample passes over the whole frame (RenderThread::process_iteratively,
// SceneManager.h:193-237), write the raw frame (FileManager::writeRawFile, FileLoading.h:166-182) and the canvas.
// Plain C++17, no HIP headers:  g++ -std=c++17 -Iinclude examples/render_glb.cpp -Lraytracing_amd -lsrx -o examples/render_glb
//
// usage: render_glb scene.glb res_x res_y subdiv spp seed out.raw [out.png]
#include <cstdio>
#include <cstdlib>
#include "spatialray_adapter.hpp"

using namespace spatialray;

int main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s scene.glb res_x res_y subdiv spp seed out.raw [out.png]\n", argv[0]); return 2; }
    try {
        const int res_x = std::atoi(argv[2]), res_y = std::atoi(argv[3]), subdiv = std::atoi(argv[4]), spp = std::atoi(argv[5]);
        const uint64_t seed = std::strtoull(argv[6], nullptr, 10);

        FileManager::GltfScene file;
        FileManager::loadGLTFFile(argv[1], file);
        file.camera.res_x = res_x; file.camera.res_y = res_y; file.camera.subdiv = subdiv;

        PackagedScene PS(file.tris);                            // Scene::package()
        PS.materials = file.materials; PS.textures = file.textures; PS.lights = file.suns;

        RayEngine RE(0, 1u << 24);
        RE.load_scene(&PS);

        FrameBuffers frame(res_x, res_y);
        Casting_Diagnostics stats;
        for (int pass = 0; pass < spp; pass++) {
            const srx_counters c = RE.trace_pass(file.camera, 0, 0, res_x, res_y, (uint32_t)pass, seed, frame.raw_output, frame.aux_output);
            stats.rays_processed += (long long)c.rays_processed;   // parent + child rays, SceneManager.h:83-90
            stats.shadows_cast += (long)c.shadows_cast;
        }
        FileManager::writeRawFile(frame.raw_output, res_x, res_y, argv[7]);
        if (argc > 8) {
            uint8_t* canvas = nullptr;
            check(srx_buffer_create((uint64_t)res_x * res_y * 4u, (void**)&canvas), "srx_buffer_create");
            ImageHandler::postProcessRaw(frame.raw_output, res_x, res_y, canvas);
            check(srx_write_png(argv[8], canvas, res_x, res_y), "srx_write_png");
            srx_buffer_destroy(canvas);
        }
        const std::vector<float> px = frame.read();
        double mean = 0;
        for (size_t i = 0; i < px.size(); i += 4) mean += px[i] + px[i + 1] + px[i + 2];
        std::printf("%zu triangles, %d x %d, %d spp: %lld rays, %ld shadow rays, mean radiance %.6f\n", file.tris.size(), res_x, res_y, spp,
                    stats.rays_processed, stats.shadows_cast, mean / (px.size() / 4 * 3));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "render_glb: %s\n", e.what());
        return 1;
    }
    return 0;
}
