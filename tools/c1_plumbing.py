#!/usr/bin/env python3
"""BASELINE.json configs[0] (C1, "plumbing, no GPU"): the Cornell-style scene at 512x512, 16 spp, rendered headless
by the CPU restatement of the reference's path (oracle/, AVX2 node and packet tests, reference RNG mode, std::thread
workers), with the numbers SURVEY 8(d) asks for. CPU only; writes one JSON object (profiles/r02_c1_cornell_cpu.json
is a run of it in the build container). Usage: python tools/c1_plumbing.py [threads] > out.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sro                      # noqa: E402
from raytracing_amd import scenes           # noqa: E402
import bench                                # noqa: E402


def main():
    model, visible, share = bench.host_cpu()
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else share
    desc = scenes.cornell(res=512, subdiv=4)
    cam = desc.camera
    t0 = time.perf_counter()
    S = sro.OracleScene(desc, build_tables=True)
    t_build = time.perf_counter() - t0
    S.set_simd(True)
    W, H = int(cam["res_x"]), int(cam["res_y"])
    acc = np.zeros((W * H, 4), np.float32)
    aux = np.zeros((W * H, 4), np.float32)
    rays = shadows = 0
    per_pass = []
    for p in range(16):
        t0 = time.perf_counter()
        st = S.render_pass(cam, (0, 0, W, H), p, 1, acc, aux if p % 2 else None, mode=1, threads=threads)
        per_pass.append(time.perf_counter() - t0)
        rays += int(st[0]); shadows += int(st[2])
    total = sum(per_pass)
    img = acc[:, :3].reshape(H, W, 3)
    out = {"config": "C1: Cornell-style box, %d triangles, 512x512, 16 spp (subdiv 4), max_generations %d, monte_carlo_max %d" %
                     (len(desc.tris), desc.max_generations, desc.monte_carlo_max),
           "path": "oracle/ (CPU restatement, AVX2 8-wide node and packet tests), reference RNG mode, interleaved rows over std::thread workers",
           "cpu_model": model, "host_cpus_visible": visible, "threads": threads,
           "scene_package_s": round(t_build, 3), "render_s": round(total, 3), "s_per_pass_median": round(float(np.median(per_pass)), 3),
           "rays": rays, "shadow_rays": shadows, "Mrays_per_s": round(rays / total / 1e6, 4),
           "image_mean_rgb": [float(x) for x in img.mean((0, 1))], "image_finite": bool(np.isfinite(img).all())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
