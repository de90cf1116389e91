"""The traversal kernel's leaf filter against the reference walk, A/B on the GPU.

Three walks over the same adversarial scenes and rays (tests/adversarial.py): the kernel with its leaf filter (box test +
normal-cone guard), the kernel with SRX_LEAF_FILTER=0 (every leaf the reference enters is tested, packet for packet)
and the CPU oracle (the reference's walk, Core.h:49-244). Hits must be bit-identical in all three; node tests equal;
packet tests: filter off == oracle, filter on == oracle minus what the oracle counts as skipped with the same rule.
"""
import numpy as np
import pytest

from adversarial import ADVERSARIAL_SCENES, adversarial_rays

pytestmark = pytest.mark.gpu


def _assert_hits_equal(g, c):
    assert np.array_equal(g["material"], c["material"])
    assert np.array_equal(g["tri_index"], c["tri_index"])
    for f in ("t", "u", "v"):
        assert np.array_equal(g[f].view(np.uint32), c[f].view(np.uint32)), f


@pytest.mark.parametrize("name", sorted(ADVERSARIAL_SCENES))
def test_filter_on_off_oracle(name, tables, monkeypatch):
    from oracle import sro
    from raytracing_amd import engine
    desc = ADVERSARIAL_SCENES[name]()
    PS = engine.package(desc, tables=tables)
    S = sro.OracleScene(desc, tables=tables)
    on, op, ot = S.packaged()
    assert np.array_equal(on, PS.avx_bvh) and np.array_equal(op, PS.avx_tri_data) and np.array_equal(ot, PS.tri_data)
    RE_on = engine.RayEngine(0)
    RE_on.load_scene(PS)
    monkeypatch.setenv("SRX_LEAF_FILTER", "0")
    RE_off = engine.RayEngine(0)
    RE_off.load_scene(PS)
    monkeypatch.delenv("SRX_LEAF_FILTER")
    sun = np.asarray(desc.suns[0][0] if desc.suns else (0.2, 0.3, 0.93), np.float32)
    skipped = 0
    for kind, (o, d) in adversarial_rays(desc, seed=5, n=40000).items():
        with np.errstate(all="ignore"):
            c, cc = S.cast_closest(o, d)
            g1, c1 = RE_on.execute_ray_cast(o, d)
            g0, c0 = RE_off.execute_ray_cast(o, d)
        _assert_hits_equal(g1, c)
        _assert_hits_equal(g0, c)
        assert int(cc[4]) == 0 and int(cc[5]) == 0, kind            # nothing the reference accepts lies in a skipped leaf
        assert int(c1[0]) == int(cc[0]) and int(c0[0]) == int(cc[0]), kind
        assert int(c0[1]) == int(cc[1]), kind                        # filter off: the reference's walk, packet for packet
        assert int(c1[1]) == int(cc[1]) - int(cc[3]), kind           # filter on: exactly the oracle's count of skipped tests
        skipped += int(cc[3])
        # shadow answers (closest hit farther than 9999, Primitives.h:44-50) from the same origins
        lit_c = S.cast_shadow(o, np.tile(sun, (len(o), 1)))
        with np.errstate(all="ignore"):
            assert np.array_equal(RE_on.execute_lighting_cast(o, sun), lit_c), kind
            assert np.array_equal(RE_off.execute_lighting_cast(o, sun), lit_c), kind
    if name in ("tilted_sheets", "atrium_8k"):
        assert skipped > 0
    RE_on.close()
    RE_off.close()
