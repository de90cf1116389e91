#!/usr/bin/env python3
"""Bounce attempts per shaded hit (the shade kernel's loop trips, Core.h:498-591) by generation cap, for a bench scene.
Usage (GPU): python tools/exp_attempts.py atrium|instanced [tris]   -> one line per max_generations 0..3 and the scene's own."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import engine, scenes

name = sys.argv[1]; tris = int(sys.argv[2]) if len(sys.argv) > 2 else (262144 if name == "atrium" else 1_000_000)
W, H = 1920, 1080
prev = None
for mg in (0, 1, 2, 3, 8):
    desc = scenes.atrium(tris, res=(W, H), subdiv=8, max_generations=mg) if name == "atrium" else \
           scenes.instanced(max(1, tris // 1000), 1000, res=(W, H), subdiv=8, max_generations=mg)
    PS = engine.package(desc)
    RE = engine.RayEngine(0); RE.load_scene(PS)
    acc = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda"); aux = torch.zeros_like(acc)
    torch.cuda.synchronize()
    c = RE.trace_pass(desc.camera, (0, 0, W, H), 0, 1, acc, aux, ray_capacity=1 << 28)
    torch.cuda.synchronize()
    cur = (c["hits"], c["pdf_lookups"], c["diffuses_cast"], c["rays_processed"])
    line = f"{name} max_generations {mg}: rays {c['rays_processed']:>10} hits {c['hits']:>10} attempts {c['pdf_lookups']:>11} children {c['diffuses_cast']:>10}  shade_ms {c['shade_ms']:.2f}"
    if prev:
        dh, da, dc = cur[0] - prev[0], cur[1] - prev[1], cur[2] - prev[2]
        if dh > 0: line += f"   | this generation: hits {dh}, attempts per bouncing hit {da / max(1, dh):.1f}, children {dc}"
    else:
        line += f"   | attempts per hit {cur[1] / max(1, cur[0]):.1f}"
    print(line, flush=True)
    prev = cur
    RE.close()
