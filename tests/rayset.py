"""The parity ray sets of SURVEY §8(d), shared by the golden generator, the CPU test and the GPU test.

Per configuration (C1 Cornell 512^2 / 16 spp, C2 atrium 262 144 tris at 1080p / 64 spp, max 8 generations): 65 536
primaries = the central 256 x 256 pixels, stratum 0, jitter from ONE sequential XorGen seeded (1, 2, 3, 4)
(XorRandom.cpp:3-69, Camera.h:91-97), and their first-bounce rays = the child rays the reference's process_ray
queues (Core.h:573-588) for those primaries in reference mode, one sequential XorGen seeded (5, 6, 7, 8).
"""
import hashlib

import numpy as np

from raytracing_amd import scenes

CONFIGS = {
    "C1_cornell": lambda: scenes.cornell(512, 4),
    "C2_atrium": lambda: scenes.atrium(262144, res=(1920, 1080), subdiv=8, max_generations=8),
}
PRIMARY_SEED = (1, 2, 3, 4)
BOUNCE_SEED = (5, 6, 7, 8)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rect_of(cam):
    W, H = int(cam["res_x"]), int(cam["res_y"])
    return ((W - 256) // 2, (H - 256) // 2, 256, 256)


def oracle_rayset(S, desc):
    """(primary origins, primary dirs, child origins, child dirs, parent index) from the oracle."""
    from oracle import sro
    cam = desc.camera
    d = sro.camera_rays_xorgen(cam, rect_of(cam), 0, PRIMARY_SEED)
    o = np.tile(np.asarray(cam["position"], np.float32), (len(d), 1))
    co, cd, par, total = sro.first_bounce_rays(S, o, d, BOUNCE_SEED)
    assert total == len(co)
    return o, d, co, cd, par


def summary(o, d, co, cd, par, hits_p, ctr_p, hits_c, ctr_c):
    """What the golden file keeps: hashes of the complete arrays + a strided subset in full."""
    sp, sc = slice(0, None, 64), slice(0, None, 509)
    return {
        "n_primary": len(d), "n_child": len(cd),
        "primary_dirs_sha256": sha(d), "child_org_sha256": sha(co), "child_dirs_sha256": sha(cd), "child_parent_sha256": sha(par),
        "primary_hits_sha256": sha(hits_p), "child_hits_sha256": sha(hits_c),
        "primary_counters": [int(ctr_p[0]), int(ctr_p[1])], "child_counters": [int(ctr_c[0]), int(ctr_c[1])],
        "primary_dirs_subset": d[sp].copy(), "primary_hits_subset": hits_p[sp].copy(),
        "child_org_subset": co[sc].copy(), "child_dirs_subset": cd[sc].copy(), "child_hits_subset": hits_c[sc].copy(),
    }
