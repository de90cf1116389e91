"""The SURVEY §8(d) parity ray sets (tests/rayset.py) on the CPU: the oracle must reproduce the committed golden files
(tests/golden/rayset_*.npz, made by tests/golden/make_rayset_golden.py), and the XorGen stream behind the primaries is
pinned independently by the pure-Python xoshiro256+ of tests/test_oracle_cpu.py."""
import os

import numpy as np
import pytest

import rayset

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(rayset.CONFIGS))
def test_oracle_reproduces_the_golden_ray_sets(name, tables):
    from oracle import sro
    g = np.load(os.path.join(GOLD, f"rayset_{name}.npz"))
    desc = rayset.CONFIGS[name]()
    S = sro.OracleScene(desc, tables=tables)
    o, d, co, cd, par = rayset.oracle_rayset(S, desc)
    assert len(d) == 65536 == int(g["n_primary"]) and len(cd) == int(g["n_child"])
    assert rayset.sha(d) == str(g["primary_dirs_sha256"])
    assert rayset.sha(co) == str(g["child_org_sha256"]) and rayset.sha(cd) == str(g["child_dirs_sha256"])
    assert rayset.sha(par) == str(g["child_parent_sha256"])
    hp, cp = S.cast_closest(o, d)
    hc, cc = S.cast_closest(co, cd)
    assert rayset.sha(hp) == str(g["primary_hits_sha256"]) and rayset.sha(hc) == str(g["child_hits_sha256"])
    assert [int(cp[0]), int(cp[1])] == list(g["primary_counters"]) and [int(cc[0]), int(cc[1])] == list(g["child_counters"])
    assert np.array_equal(d[::64].view(np.uint32), g["primary_dirs_subset"].view(np.uint32))
    assert np.array_equal(hp[::64], g["primary_hits_subset"]) and np.array_equal(hc[::509], g["child_hits_subset"])
    # the reference-mode filter never lets an accept fall into a leaf the product skips
    assert int(cp[4]) == 0 and int(cc[4]) == 0
    assert (hp["material"] != 0xFFFFFFFF).mean() > 0.9


def test_primary_jitter_is_the_xorgen_1234_stream():
    """The first pixel's jitter must be the first two scalar draws of XorGen(1, 2, 3, 4) after its 16 seeding draws:
    ties the ray set to the generator the golden vectors and the pure-Python xoshiro pin."""
    from oracle import sro
    from raytracing_amd import scenes
    desc = rayset.CONFIGS["C1_cornell"]()
    cam = desc.camera
    x0, y0, _, _ = rayset.rect_of(cam)
    sc, _, _ = sro.xorgen_stream(rayset.PRIMARY_SEED, 2, 0)
    d = sro.camera_rays_xorgen(cam, (x0, y0, 1, 1), 0, rayset.PRIMARY_SEED)[0].astype(np.float64)
    # invert Camera.h:91-97 for an unrotated camera: lens point = focal * d.xz / d.y
    f = float(cam["focal_position"][1])
    lens_x, lens_y = f * d[0] / d[1], f * d[2] / d[1]
    pw, ph = float(cam["lens_width"]) / int(cam["res_x"]), float(cam["lens_height"]) / int(cam["res_y"])
    sw, sh = pw / int(cam["subdiv"]), ph / int(cam["subdiv"])
    jx = lens_x - (-float(cam["lens_width"]) / 2 + pw * x0)
    jy = lens_y - (-float(cam["lens_height"]) / 2 + ph * y0)
    assert abs(jx - sc[0] * sw) < 3e-7 and abs(jy - sc[1] * sh) < 3e-7        # float32 directions: ~1e-7 on a lens coordinate of 0.5; the stratum is 4.9e-4 wide
