#!/usr/bin/env python3
"""Experiment: does ordering secondary rays (direction octant / origin cell) speed up the closest-hit kernel?
Secondary rays = 8 random hemisphere directions per primary hit of the bench scene (children of one parent are
consecutive, as the shade kernel appends them)."""
import ctypes as C, time, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import engine, scenes, _lib

W, H = 1920, 1080
desc = scenes.atrium(262144, res=(W, H), subdiv=8, max_generations=8)
PS = engine.package(desc)
RE = engine.RayEngine(0); RE.load_scene(PS)
L = _lib.lib()
d, k = engine.camera_rays(desc.camera, (0, 0, W, H), 0, 1)
cam = desc.camera
org = np.broadcast_to(np.asarray(cam["position"], np.float32), d.shape).copy()
if org is None:
    print(cam.dtype.names); sys.exit(1)

def cast(o, dd, reps=3):
    n = o.shape[0]
    hits = torch.zeros(n * 5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = L.srx_cast_closest_device(RE.scene, C.c_void_p(o.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_uint32(n), C.c_void_p(hits.data_ptr()), None, None)
        assert rc == 0
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return hits.view(n, 5), best

o_t = torch.from_numpy(org).cuda(); d_t = torch.from_numpy(d).cuda()
hits, t = cast(o_t, d_t)
print(f"primary: {o_t.shape[0] / t / 1e6:.0f} Mrays/s")
tt = hits[:, 1].view(torch.float32)
ok = hits[:, 0] != -1
pos = (o_t + d_t * tt[:, None] * 0.999)[ok]
tri = hits[:, 4][ok].long()
P = torch.from_numpy(np.ascontiguousarray(desc.tris["p"])).cuda()[tri]
nrm = torch.linalg.cross(P[:, 3:6] - P[:, 0:3], P[:, 6:9] - P[:, 0:3])
nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-20)
nrm = torch.where(((nrm * d_t[ok]).sum(1) > 0)[:, None], -nrm, nrm)
K = 8
g = torch.Generator(device="cuda").manual_seed(5)
pos = pos.repeat_interleave(K, dim=0); nrm = nrm.repeat_interleave(K, dim=0)
r = torch.randn(pos.shape, device="cuda", generator=g); r = r / r.norm(dim=1, keepdim=True)
dirs = nrm + r * 0.999; dirs = dirs / dirs.norm(dim=1, keepdim=True)      # cosine-weighted hemisphere
pos = (pos + nrm * 1e-4).contiguous(); dirs = dirs.contiguous()
n = pos.shape[0]
_, t0 = cast(pos, dirs)
print(f"natural order ({n} rays): {n / t0 / 1e6:.0f} Mrays/s")
octant = ((dirs[:, 0] < 0).long() | ((dirs[:, 1] < 0).long() << 1) | ((dirs[:, 2] < 0).long() << 2))
perm = torch.sort(octant, stable=True)[1]
_, t1 = cast(pos[perm].contiguous(), dirs[perm].contiguous())
print(f"octant-sorted (stable): {n / t1 / 1e6:.0f} Mrays/s")
# octant + coarse origin cell (Morton of 6 bits per axis)
lo, hi = pos.min(0)[0], pos.max(0)[0]
q = ((pos - lo) / (hi - lo).clamp_min(1e-9) * 63.999).long()
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
mort = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
perm2 = torch.sort((octant << 18) | mort, stable=True)[1]
_, t2 = cast(pos[perm2].contiguous(), dirs[perm2].contiguous())
print(f"octant+morton-sorted: {n / t2 / 1e6:.0f} Mrays/s")
perm3 = torch.sort((mort << 3) | octant, stable=True)[1]
_, t3 = cast(pos[perm3].contiguous(), dirs[perm3].contiguous())
print(f"morton+octant-sorted: {n / t3 / 1e6:.0f} Mrays/s")
perm4 = torch.randperm(n, device="cuda")
_, t4 = cast(pos[perm4].contiguous(), dirs[perm4].contiguous())
print(f"random order: {n / t4 / 1e6:.0f} Mrays/s")
