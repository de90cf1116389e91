"""Generates the golden fixtures in tests/golden/ from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`). The reference has no tests, golden vectors or fixtures for this path
and its Core.h stack cannot be built here, so these vectors pin the oracle against ITSELF over time
(regression) — they are not outputs of the reference. Independent pins live in tests/test_oracle_cpu.py
(pure-Python xoshiro256+/splitmix64, libm tables, and the reference's own XYZ.cpp built in oracle/_ref).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sro                      # noqa: E402
from raytracing_amd import scenes           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    g = {}
    sc, po, raw = sro.xorgen_stream((1, 2, 3, 4), 16, 4)
    g["xorgen_1234_scalars_bits"] = sc.view(np.uint32).tolist()
    g["xorgen_1234_pools_bits"] = po.view(np.uint32).reshape(-1).tolist()
    g["xorgen_1234_raw_after"] = [int(x) for x in raw]
    x = np.linspace(0.0, 2 * np.pi, 33, dtype=np.float32)
    s, c, t = sro.trig(x)
    g["trig_x_bits"] = x.view(np.uint32).tolist()
    g["full_sin_bits"] = s.view(np.uint32).tolist()
    g["full_cos_bits"] = c.view(np.uint32).tolist()
    g["poly_acos_bits"] = sro.poly_acos(np.linspace(-1, 1, 17, dtype=np.float32)).view(np.uint32).tolist()
    po, scl, ck = sro.keyed_stream(0x0123456789ABCDEF, 2, 4)
    g["keyed_pools_bits"] = po.view(np.uint32).reshape(-1).tolist()
    g["keyed_scalars_bits"] = scl.view(np.uint32).tolist()
    g["keyed_child_key"] = int(ck)

    desc = scenes.cornell(32, 2, with_sun=True)
    S = sro.OracleScene(desc)
    a, p = S.tables()
    g["arccos_sha256"] = sha(a)
    g["pdf_sha256"] = sha(p)
    g["pdf_spot"] = {str(i): int(p[i].view(np.uint32)) for i in (0, 1, 1023, 1024, 512 * 1024 + 512, 1024 * 1024 - 1)}
    n, pk, tr = S.packaged()
    g["cornell_counts"] = [len(n), len(pk), len(tr)]
    g["cornell_nodes_sha256"], g["cornell_packets_sha256"], g["cornell_tris_sha256"] = sha(n), sha(pk), sha(tr)
    cam = desc.camera
    dirs, keys = sro.camera_rays_keyed(cam, (0, 0, 32, 32), 0, 99)
    org = np.tile(np.asarray(cam["position"], np.float32), (len(dirs), 1))
    hits, ctr = S.cast_closest(org, dirs)
    np.savez_compressed(os.path.join(OUT, "cornell_primary_hits.npz"), dirs=dirs, keys=keys, hits=hits, counters=ctr)
    rad, st = S.process_rays_keyed(org[:256], dirs[:256], keys[:256], 0)
    np.savez_compressed(os.path.join(OUT, "cornell_process_ray.npz"), radiance=rad, stats=st)

    d2 = scenes.soup(1500, seed=3)
    S2 = sro.OracleScene(d2, tables=(a, p))
    n2, pk2, tr2 = S2.packaged()
    g["soup_counts"] = [len(n2), len(pk2), len(tr2)]
    g["soup_nodes_sha256"], g["soup_packets_sha256"], g["soup_tris_sha256"] = sha(n2), sha(pk2), sha(tr2)
    rng = np.random.default_rng(5)
    o = rng.uniform(-4, 4, (2000, 3)).astype(np.float32)
    dd = rng.normal(size=(2000, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    h2, c2 = S2.cast_closest(o, dd.astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "soup_hits.npz"), org=o, dirs=dd.astype(np.float32), hits=h2, counters=c2)
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
