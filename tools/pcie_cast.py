#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer cast entry point (srx_cast_closest: copy rays in, cast, copy hits out)
beside the device-pointer one (srx_cast_closest_device) on the same rays of the bench scene."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import engine, scenes, _lib

W, H = 1920, 1080
desc = scenes.atrium(262144, res=(W, H), subdiv=8, max_generations=8)
RE = engine.RayEngine(0); RE.load_scene(engine.package(desc))
L = _lib.lib()
d, _ = engine.camera_rays(desc.camera, (0, 0, W, H), 0, 1)
d = np.tile(d, (8, 1)); n = len(d)
o = np.tile(np.asarray(desc.camera["position"], np.float32).reshape(1, 3), (n, 1))
for _ in range(2):
    t0 = time.perf_counter(); RE.execute_ray_cast(o, d); t_host = time.perf_counter() - t0
od, dd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
hits = torch.zeros(n * 5, dtype=torch.int32, device="cuda")
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert L.srx_cast_closest_device(RE.scene, C.c_void_p(od.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_uint32(n), C.c_void_p(hits.data_ptr()), None, None) == 0
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{n} primary rays: host pointers {n / t_host / 1e6:.0f} Mrays/s ({t_host * 1e3:.1f} ms, {n * 44 / t_host / 1e9:.1f} GB/s over PCIe incl. allocation); "
      f"device pointers {n / best / 1e6:.0f} Mrays/s ({best * 1e3:.1f} ms)")
