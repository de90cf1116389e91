"""The SURVEY §8(d) parity ray sets on the GPU (rows a4-a7, a15, a17): 65 536 primaries of the C1 and C2 cameras with
jitter from the reference's XorGen seeded (1, 2, 3, 4) — generated ON THE DEVICE by srx_camera_rays_xorgen and required
bit-identical to the oracle's and to the committed golden hashes — and their first-bounce rays captured from the oracle's
reference-mode process_ray (Core.h:573-588). Both sets are cast on the GPU: tri_index, material, t, u, v at 0 ulp and
the traversal counters against the oracle, and the hit arrays against the golden hashes."""
import os

import numpy as np
import pytest

import rayset

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_device_xorgen_streams(tables):
    """XorGen on the device (XorRandom.cpp:3-69, 119-132): pools and scalars of three seeds equal the oracle's, and seed
    (1, 2, 3, 4) equals the committed golden vectors."""
    import json
    from oracle import sro
    from raytracing_amd import engine
    engine.RayEngine(0)
    for seed in ((1, 2, 3, 4), (5, 6, 7, 8), (0xDEADBEEF, 2**63 + 5, 7, 2**64 - 1)):
        po, sc = engine.xorgen_stream(seed, 300, 1000)
        sc_o, po_o, _ = sro.xorgen_stream(seed, 1000, 300)
        assert np.array_equal(po.view(np.uint32), po_o.view(np.uint32))
        assert np.array_equal(sc.view(np.uint32), sc_o.view(np.uint32))
    with open(os.path.join(GOLD, "golden.json")) as fh:
        g = json.load(fh)
    po, sc = engine.xorgen_stream((1, 2, 3, 4), 4, 16)
    assert po.view(np.uint32).reshape(-1).tolist() == g["xorgen_1234_pools_bits"]
    assert sc.view(np.uint32).tolist() == g["xorgen_1234_scalars_bits"]


@pytest.mark.parametrize("name", sorted(rayset.CONFIGS))
def test_ray_sets_cast_bit_exact(name, tables):
    from oracle import sro
    from raytracing_amd import engine
    g = np.load(os.path.join(GOLD, f"rayset_{name}.npz"))
    desc = rayset.CONFIGS[name]()
    PS = engine.package(desc, tables=tables)
    S = sro.OracleScene(desc, tables=tables)
    on, op, ot = S.packaged()
    assert np.array_equal(on, PS.avx_bvh) and np.array_equal(op, PS.avx_tri_data) and np.array_equal(ot, PS.tri_data)
    RE = engine.RayEngine(0)
    RE.load_scene(PS)
    cam = desc.camera
    o, d, co, cd, par = rayset.oracle_rayset(S, desc)
    # primaries from the device's XorGen
    dg = engine.camera_rays_xorgen(cam, rayset.rect_of(cam), 0, rayset.PRIMARY_SEED)
    assert np.array_equal(dg.view(np.uint32), d.view(np.uint32))
    assert rayset.sha(dg) == str(g["primary_dirs_sha256"])
    assert rayset.sha(co) == str(g["child_org_sha256"]) and rayset.sha(cd) == str(g["child_dirs_sha256"])
    for (ro, rd, key) in ((o, dg, "primary"), (co, cd, "child")):
        hg, cg, adv_g = RE.execute_ray_cast(ro, rd, want_advanced=True)
        hc, cc, adv_c = S.cast_closest(ro, rd, want_advanced=True)
        assert np.array_equal(hg["material"], hc["material"]) and np.array_equal(hg["tri_index"], hc["tri_index"]), key
        for f in ("t", "u", "v"):
            assert np.array_equal(hg[f].view(np.uint32), hc[f].view(np.uint32)), (key, f)
        assert np.array_equal(adv_g.view(np.uint32), adv_c.view(np.uint32)), key              # Core.h:327
        assert int(cg[0]) == int(cc[0]) and int(cg[1]) == int(cc[1]) - int(cc[3]), key
        assert rayset.sha(hg) == str(g[f"{key}_hits_sha256"]), key
        assert int(cc[4]) == 0, key
    RE.close()
