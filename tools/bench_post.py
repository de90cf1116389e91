#!/usr/bin/env python3
"""Times the output-stage kernel (srx_post_process) on 4K and 1080p frames. Run under
`rocprofv3 --kernel-trace --stats` for the kernel's own duration; the wall time printed here includes the call's
launch + device synchronisation. Algorithmic bytes per pixel: 16 in (float4 framebuffer) + 4 out (RGBA8)."""
import json, sys, time
import torch
from raytracing_amd import output

res = {}
for name, (rx, ry) in {"4k": (3840, 2160), "1080p": (1920, 1080)}.items():
    acc = torch.rand((rx * ry, 4), device="cuda") * 2
    for _ in range(3):
        output.post_process_raw(acc, rx, ry)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        output.post_process_raw(acc, rx, ry)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res[name] = {"pixels": rx * ry, "wall_us_per_call": dt * 1e6, "algorithmic_bytes": rx * ry * 20,
                 "wall_GBps": rx * ry * 20 / dt / 1e9}
print(json.dumps(res))
