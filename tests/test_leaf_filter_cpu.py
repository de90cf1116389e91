"""Soundness of the product's leaf filter, checked on the CPU against the reference walk (oracle).

The traversal kernel skips a leaf when (1) the ray misses the padded tight box of its triangles and (2) the ray is not
close to the plane of any of its triangles (normal-cone guard). The oracle keeps the reference's walk
(Core.h:49-244: every leaf whose origin-grown box the ray enters is tested) and evaluates the same two conditions with
the same arithmetic, so it can count what must never happen: the reference's fp32 triangle test
(Primitives.h:159-202) taking a closest-hit update, or even passing a lane with t > 0, inside a leaf the filter skips.
Adversarial inputs (tests/adversarial.py): edge-on rays with tilts down to 1e-9 rad, slivers of aspect up to 1e6,
|coordinates| 1e4 with 1e-3 features, rays along box faces, rays that start on triangle planes, far origins.
Without condition (2) these sets DO produce such accepts (tools/exp_leaf_filter.py history in DESIGN.md section 5).
"""
import numpy as np
import pytest

from adversarial import ADVERSARIAL_SCENES, adversarial_rays


@pytest.mark.parametrize("name", sorted(ADVERSARIAL_SCENES))
def test_no_reference_accept_inside_a_skipped_leaf(name):
    from oracle import sro
    desc = ADVERSARIAL_SCENES[name]()
    S = sro.OracleScene(desc, build_tables=False)
    saved = 0
    for seed in (5, 6):
        for kind, (o, d) in adversarial_rays(desc, seed=seed, n=40000).items():
            with np.errstate(all="ignore"):
                _, ctr = S.cast_closest(o, d)
            assert int(ctr[4]) == 0, f"{name}/{kind}: closest-hit update inside a skipped leaf"
            assert int(ctr[5]) == 0, f"{name}/{kind}: triangle test passed inside a skipped leaf"
            saved += int(ctr[3])
    if name in ("tilted_sheets", "atrium_8k"):
        assert saved > 0, "the filter never skipped anything: the test does not exercise it"


def test_filter_switch_off_counts_nothing():
    from oracle import sro
    desc = ADVERSARIAL_SCENES["atrium_8k"]()
    S = sro.OracleScene(desc, build_tables=False)
    o, d = adversarial_rays(desc, seed=5, n=8000)["incoherent"]
    h1, c1 = S.cast_closest(o, d)
    S.set_leaf_filter(False)
    h0, c0 = S.cast_closest(o, d)
    assert np.array_equal(h0, h1)                       # the oracle's hits never depend on the filter
    assert int(c1[3]) > 0 and int(c0[3]) == 0 and int(c0[6]) == 0
    assert int(c0[0]) == int(c1[0]) and int(c0[1]) == int(c1[1])


def test_observed_margin_of_the_soundness_bound(capsys):
    """A number on the filter's margin. DESIGN.md section 5 argues that a lane the reference's fp32
    test PASSES, in a leaf whose guard condition (2) holds for the ray, has its triangle point within 1.6e-4 |o - p1| of
    its ray point, hence the ray point within 2.7e-4 D of the leaf's tight box, against a pad of 5e-4 D. Here both are
    MEASURED over the adversarial scenes and ray sets (in double, from the fp32 t, u, v of the reference's arithmetic):
    the worst observed value must stay below the bound with a factor of 2 to spare, and is printed."""
    from oracle import sro
    worst_r, worst_o, lanes = 0.0, 0.0, 0
    rows = []
    for name in sorted(ADVERSARIAL_SCENES):
        desc = ADVERSARIAL_SCENES[name]()
        S = sro.OracleScene(desc, build_tables=False)
        wr, wo, ln = 0.0, 0.0, 0
        for kind, (o, d) in adversarial_rays(desc, seed=5, n=40000).items():
            with np.errstate(all="ignore"):
                r, out, n = S.filter_margins(o, d)
            wr, wo, ln = max(wr, r), max(wo, out), ln + n
        rows.append((name, wr, wo, ln))
        worst_r, worst_o, lanes = max(worst_r, wr), max(worst_o, wo), lanes + ln
    with capsys.disabled():
        for name, wr, wo, ln in rows:
            print(f"\n  leaf filter margin {name:>16}: max |r|/|T| = {wr:.3e} (bound 1.6e-4), max outside/D = {wo:.3e} (pad 5e-4), {ln} passing lanes", end="")
        print(f"\n  overall: |r|/|T| uses {worst_r / 1.6e-4:.3f} of its bound, outside/D uses {worst_o / 5e-4:.3f} of the pad")
    assert lanes > 50_000
    assert worst_r < 0.5 * 1.6e-4, worst_r
    assert worst_o < 0.5 * 5e-4, worst_o
