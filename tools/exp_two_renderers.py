#!/usr/bin/env python3
"""Experiment: two renderers (own queues and streams) rendering different passes of the same frame at the same time
from two host threads, against one renderer doing the same passes one after the other. Tells whether keeping two
passes in flight (shade of one beside the traversal of the other) would pay."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import engine, scenes

W, H = 1920, 1080
desc = scenes.atrium(262144, res=(W, H), subdiv=8, max_generations=8)
PS = engine.package(desc)
cap = 1 << 28
REs = [engine.RayEngine(0) for _ in range(2)]
bufs = []
for RE in REs:
    RE.load_scene(PS)
    RE.set_timing(False, ray_capacity=cap)
    bufs.append((torch.zeros((W * H, 4), device="cuda"), torch.zeros((W * H, 4), device="cuda")))

def run(i, passes, out):
    rays = 0
    for p in passes:
        c = REs[i].trace_pass(desc.camera, (0, 0, W, H), p, 1, bufs[i][0], bufs[i][1], ray_capacity=cap)
        rays += int(c["rays_processed"])
    out[i] = rays

for rep in range(2):
    out = [0, 0]
    run(0, [0], out); run(1, [1], out)                       # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(0, [2, 3, 4, 5], out)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"one renderer, 4 passes: {out[0] / t1 / 1e6:.1f} Mrays/s ({t1 * 1e3 / 4:.2f} ms per pass)")
    out = [0, 0]
    th = [threading.Thread(target=run, args=(i, [6 + 2 * i, 7 + 2 * i], out)) for i in range(2)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print(f"two renderers, 2 passes each at once: {sum(out) / t2 / 1e6:.1f} Mrays/s ({t2 * 1e3 / 4:.2f} ms per pass)")
