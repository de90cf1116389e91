#!/usr/bin/env python3
"""Cast-only timing of the closest-hit kernel on a fixed set of secondary rays of the bench scene (8 hemisphere
directions per primary hit). First run (the stock library) writes the rays to /tmp/srx_rays.pt, later runs (SRX_LIB =
a variant build) load them, so variants that return unusable hit records can still be timed."""
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

def cast(o, dd, reps=5):
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

F = "/tmp/srx_rays.pt"
if os.path.exists(F):
    pos, dirs = torch.load(F, weights_only=True)
    pos, dirs = pos.cuda(), dirs.cuda()
else:
    d, k = engine.camera_rays(desc.camera, (0, 0, W, H), 0, 1)
    org = np.broadcast_to(np.asarray(desc.camera["position"], np.float32), d.shape).copy()
    o_t = torch.from_numpy(org).cuda(); d_t = torch.from_numpy(d).cuda()
    hits, t = cast(o_t, d_t)
    tt = hits[:, 1].view(torch.float32)
    ok = hits[:, 0] != -1
    pos = (o_t + d_t * tt[:, None] * 0.999)[ok]
    tri = hits[:, 4][ok].long()
    P = torch.from_numpy(np.ascontiguousarray(desc.tris["p"])).cuda()[tri]
    nrm = torch.linalg.cross(P[:, 3:6] - P[:, 0:3], P[:, 6:9] - P[:, 0:3])
    nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-20)
    nrm = torch.where(((nrm * d_t[ok]).sum(1) > 0)[:, None], -nrm, nrm)
    g = torch.Generator(device="cuda").manual_seed(5)
    pos = pos.repeat_interleave(8, dim=0); nrm = nrm.repeat_interleave(8, dim=0)
    r = torch.randn(pos.shape, device="cuda", generator=g); r = r / r.norm(dim=1, keepdim=True)
    dirs = nrm + r * 0.999; dirs = dirs / dirs.norm(dim=1, keepdim=True)
    pos = (pos + nrm * 1e-4).contiguous(); dirs = dirs.contiguous()
    torch.save((pos.cpu(), dirs.cpu()), F)
n = pos.shape[0]
_, t0 = cast(pos, dirs)
print(f"{os.environ.get('SRX_LIB', 'libsrx.so').split('/')[-1]}: {n} secondary rays: {n / t0 / 1e6:.0f} Mrays/s ({t0 * 1e3:.2f} ms)")
