#!/usr/bin/env python3
"""Experiment: what do the reference's origin-inflated leaf boxes (quirk Q1, BVH.h:27-34) cost? Same tree, leaf child
boxes replaced by the padded tight bounds of their triangles; compares Mrays/s and tests per ray on the bench frame."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import engine, scenes

W, H = 1920, 1080
desc = scenes.atrium(262144, res=(W, H), subdiv=8, max_generations=8)
PS = engine.package(desc)

def tighten(PS):
    n = PS.avx_bvh.copy()
    pk = PS.avx_tri_data
    f = lambda a, off: a[:, off:off + 32].copy().view(np.float32)            # [n, 8]
    p1 = np.stack([f(pk, 0), f(pk, 32), f(pk, 64)], -1)                      # [np, 8, 3]
    e1 = np.stack([f(pk, 96), f(pk, 128), f(pk, 160)], -1)
    e2 = np.stack([f(pk, 192), f(pk, 224), f(pk, 256)], -1)
    cnt = pk[:, 420].view(np.int8).astype(int)
    verts = np.stack([p1, p1 + e1, p1 + e2], 2)                              # [np, 8, 3 verts, 3]
    lane_ok = np.arange(8)[None, :] < cnt[:, None]
    big = np.float32(3e38)
    vmin = np.where(lane_ok[:, :, None, None], verts, big).min((1, 2))       # [np, 3]
    vmax = np.where(lane_ok[:, :, None, None], verts, -big).max((1, 2))
    leaf = n[:, 192:224].copy().view(np.uint32); idx = n[:, 224:256].copy().view(np.uint32)
    nv = n.view(np.float32).reshape(len(n), -1)
    for i, k in zip(*np.nonzero(leaf)):
        a, b = idx[i, k], idx[i, k] + leaf[i, k]
        mn, mx = vmin[a:b].min(0), vmax[a:b].max(0)
        pad = (mx - mn) * 1e-4 + 1e-5
        mn, mx = mn - pad, mx + pad
        for ax in range(3):
            nv[i, (0 + 32 * ax) // 4 + k] = mx[ax]
            nv[i, (96 + 32 * ax) // 4 + k] = mn[ax]
    out = engine.PackagedScene(n, PS.avx_tri_data, PS.tri_data, PS.materials, PS.textures, PS.lights, PS.max_generations,
                               PS.monte_carlo_max, PS.monte_carlo_modifier, PS.tables)
    return out

def run(PS, label):
    RE = engine.RayEngine(0); RE.load_scene(PS); RE.set_timing(True, ray_capacity=1 << 27)
    acc = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda"); aux = torch.zeros_like(acc)
    RE.trace_pass(desc.camera, (0, 0, W, H), 0, 1, acc, aux, ray_capacity=1 << 27)
    torch.cuda.synchronize(); t0 = time.perf_counter(); tot = {}
    for p in range(1, 4):
        c = RE.trace_pass(desc.camera, (0, 0, W, H), p, 1, acc, aux, ray_capacity=1 << 27)
        for k, v in c.items(): tot[k] = tot.get(k, 0) + v
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = tot["rays_processed"]
    print(f"{label}: {r / dt / 1e6:.0f} Mrays/s, nodes/ray {tot['nodes'] / r:.2f}, packets/ray {tot['packets'] / r:.2f}, "
          f"shadow nodes/ray {tot['shadow_nodes'] / tot['shadows_cast']:.2f} packets {tot['shadow_packets'] / tot['shadows_cast']:.2f}, "
          f"trace {tot['trace_ms']:.1f} shadow {tot['shadow_ms']:.1f} shade {tot['shade_ms']:.1f} ms, mean {acc[:, :3].mean().item():.5f}")
    RE.close()

run(PS, "reference leaf boxes (Q1)")
run(tighten(PS), "tight leaf boxes      ")
