// examples/render_glb.cpp — a headless render loop written against the reference's names, over the C-ABI.
//
// What the reference's Main.cpp:49-205 + SceneManager::render do for one frame, without the window: load a .glb
// (FileManager::loadGLTFFile, FileLoading.h:200-224), package it (Scene::package, Scene.h:90-179), hand it to the engine
// (RayEngine::load_scene, Core.h:46-48), take `spp` sample passes over the whole frame (RenderThread::process_iteratively,
// SceneManager.h:193-237), write the raw frame (FileManager::writeRawFile, FileLoading.h:166-182) and the canvas.
// Plain C++17, no HIP headers:  g++ -std=c++17 -Iinclude examples/render_glb.cpp -Lraytracing_amd -lsrx -o examples/render_glb
//
// usage: render_glb scene.glb res_x res_y subdiv spp seed out.raw [out.png] [--rank R --nranks N --id-file F [--device D]]
// With --id-file the process is one rank of a multi-GPU render (one process per GPU): it renders its tile rows and the
// frame is gathered through spatialray::Group (srx_group, RCCL); rank 0 writes the files.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <thread>
#include <sys/stat.h>
#include "spatialray_adapter.hpp"

using namespace spatialray;

int main(int argc, char** argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s scene.glb res_x res_y subdiv spp seed out.raw [out.png] [--rank R --nranks N --id-file F [--device D]]\n"
                             "  one process per GPU: every rank renders its tile rows (4-row groups dealt round-robin), then the frame is gathered\n"
                             "  over RCCL; rank 0 writes the communicator id to F, the other ranks read it from there; only rank 0 writes files\n", argv[0]);
        return 2;
    }
    try {
        int rank = 0, nranks = 1, device = 0;
        std::string id_file;
        int n_pos = argc;
        for (int i = 8; i < argc; i++) {
            const std::string a = argv[i];
            if (a.rfind("--", 0) == 0 && n_pos == argc) n_pos = i;
            if (a == "--rank" && i + 1 < argc) rank = std::atoi(argv[++i]);
            else if (a == "--nranks" && i + 1 < argc) nranks = std::atoi(argv[++i]);
            else if (a == "--id-file" && i + 1 < argc) id_file = argv[++i];
            else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        }
        const bool grouped = !id_file.empty();
        if (grouped && (nranks < 1 || rank < 0 || rank >= nranks)) throw std::runtime_error("bad --rank / --nranks");
        // An id file left by an earlier job that died would make the other ranks join a dead communicator and hang in
        // ncclCommInitRank: rank 0 removes it before anything else and again once every rank has joined, and the other
        // ranks take no file older than 10 minutes before their own start (start the ranks together, rank 0 first or not).
        const time_t started = time(nullptr);
        if (grouped && rank == 0) std::remove(id_file.c_str());
        argc = n_pos;
        const int res_x = std::atoi(argv[2]), res_y = std::atoi(argv[3]), subdiv = std::atoi(argv[4]), spp = std::atoi(argv[5]);
        const uint64_t seed = std::strtoull(argv[6], nullptr, 10);

        FileManager::GltfScene file;
        FileManager::loadGLTFFile(argv[1], file);
        file.camera.res_x = res_x; file.camera.res_y = res_y; file.camera.subdiv = subdiv;

        PackagedScene PS(file.tris);                            // Scene::package()
        PS.materials = file.materials; PS.textures = file.textures; PS.lights = file.suns;

        RayEngine RE(device, 1u << 24);
        RE.load_scene(&PS);
        // multi-GPU: the communicator (RayEngine's constructor made `device` current)
        std::unique_ptr<Group> group;
        if (grouped) {
            std::vector<char> id(128);
            if (rank == 0) {
                id = Group::unique_id();
                std::ofstream f(id_file + ".tmp", std::ios::binary);
                f.write(id.data(), 128);
                f.close();
                std::rename((id_file + ".tmp").c_str(), id_file.c_str());
            } else {
                for (int tries = 0;; tries++) {
                    struct stat sb;
                    const bool fresh = stat(id_file.c_str(), &sb) == 0 && sb.st_mtime + 600 >= started;
                    std::ifstream f(id_file, std::ios::binary);
                    if (fresh && f && f.read(id.data(), 128)) break;
                    if (tries > 600) throw std::runtime_error("no communicator id appeared in " + id_file);
                    std::this_thread::sleep_for(std::chrono::milliseconds(100));
                }
            }
            group.reset(new Group(id, rank, nranks));
            if (rank == 0) std::remove(id_file.c_str());         // every rank has joined: the id has served
        }

        FrameBuffers frame(res_x, res_y);
        Casting_Diagnostics stats;
        for (int pass = 0; pass < spp; pass++) {
            // this rank's rows of the pass (all of them without a group): SceneManager::assign_iterative_groups, SceneManager.h:441-459
            const srx_counters c = RE.trace_pass_interleaved(file.camera, 0, 0, res_x, res_y, 4, rank, nranks, (uint32_t)pass, seed, frame.raw_output,
                                                             frame.aux_output);
            stats.rays_processed += (long long)c.rays_processed;   // parent + child rays, SceneManager.h:83-90
            stats.shadows_cast += (long)c.shadows_cast;
        }
        if (group) {
            group->gather_framebuffer(frame.raw_output, frame.aux_output, res_x, res_y, 4);     // SceneManager::get_screen_update, :801-815
            // the job's ray counts: SceneManager::gather_thread_stats (:488-502) sums the threads' counters
            double h_sums[2] = {(double)stats.rays_processed, (double)stats.shadows_cast};
            double* d_sums = nullptr;
            check(srx_buffer_create(sizeof(h_sums), (void**)&d_sums), "srx_buffer_create");
            check(srx_buffer_write(d_sums, sizeof(h_sums), h_sums), "srx_buffer_write");
            group->allreduce(d_sums, 2, nullptr, 0);
            check(srx_buffer_read(d_sums, sizeof(h_sums), h_sums), "srx_buffer_read");
            srx_buffer_destroy(d_sums);
            stats.rays_processed = (long long)h_sums[0]; stats.shadows_cast = (long)h_sums[1];
            if (rank != 0) return 0;
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
