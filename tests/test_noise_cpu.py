"""CPU tests of the adaptive-sampling controller restatement (oracle/sro_noise.h; SceneManager.h:526-692, :193-198).
The reference holds no fixture for it (parity unpinned upstream): the checks are the formulas re-derived in numpy,
structural invariants of the region list and a committed golden region list (tests/golden/noise_golden.json)."""
import json
import os

import numpy as np
import pytest

from oracle import sro
from noise_inputs import framebuffers

HERE = os.path.dirname(os.path.abspath(__file__))


def test_error_map_formula():
    acc, aux = framebuffers(64, 40)
    err, mn = sro.noise_error_map(acc, aux)
    lum = lambda c: (c[:, 0] * np.float32(0.2126) + c[:, 1] * np.float32(0.7152)) + c[:, 2] * np.float32(0.0722)
    L, La = lum(acc), lum(aux)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = np.abs(L - La) / np.maximum(np.power(L, np.float32(1.1), dtype=np.float32), L)
    ref = np.where(L == 0, np.float32(0), ref).astype(np.float32)
    assert np.allclose(err, ref, rtol=2e-6, atol=0)
    assert (err[L == 0] == 0).all() and (L == 0).sum() == 45
    assert mn == 68


def _partition_ok(rects, W, H):
    cover = np.zeros((H, W), np.int32)
    for x0, y0, x1, y1 in rects:
        assert 0 <= x0 <= x1 <= W and 0 <= y0 <= y1 <= H
        cover[y0:y1, x0:x1] += 1
    return (cover == 1).all()


def test_regions_partition_the_frame_and_split_only_after_minimum_samples():
    W, H = 96, 64
    acc, aux = framebuffers(W, H)
    err, mn = sro.noise_error_map(acc, aux)
    N = sro.OracleNoise(W, H, seed=5)
    N.analyze(err, 10)                                   # below minimum_samples (64): statistics only (:582)
    rects, stats = N.regions()
    assert len(rects) == 1 and rects[0].tolist() == [0, 0, W, H]
    assert np.isclose(stats[0, 1], err.astype(np.float64).mean(), rtol=1e-6)
    pr, wrote = N.priorities(10)
    assert not wrote and (pr == 1).all()                 # :669 early return keeps block::priority at 1 (:36)
    N.analyze(err, mn)
    rects, stats = N.regions()
    assert len(rects) > 8 and _partition_ok(rects, W, H)
    w, h = rects[:, 2] - rects[:, 0], rects[:, 3] - rects[:, 1]
    assert (w >= 8).all() and (h >= 8).all()             # minimum_size (:527, :613, :632)
    N2 = sro.OracleNoise(W, H, seed=5)
    N2.analyze(err, 10); N2.analyze(err, mn)
    r2, s2 = N2.regions()
    assert np.array_equal(rects, r2) and np.array_equal(stats, s2)   # deterministic
    pr, wrote = N.priorities(mn)
    assert wrote and pr.shape == ((W // 4) * (H // 4),) and (pr >= 0).all() and (pr <= 1.0 + 1e-6).all()
    noisy = pr.reshape(H // 4, W // 4)[:, : W // 5 // 4].mean()
    quiet = pr.reshape(H // 4, W // 4)[: H // 8, -4:].mean()
    assert noisy > 10 * quiet                            # the noisy side gets the samples


def test_converged_small_regions_go_inactive():
    W, H = 32, 16
    acc, aux = framebuffers(W, H)
    aux[:, :3] = acc[:, :3]                              # zero error everywhere
    err, mn = sro.noise_error_map(acc, aux)
    assert (err == 0).all()
    N = sro.OracleNoise(W, H, seed=1)
    N.analyze(err, mn)
    rects, stats = N.regions()
    assert len(rects) == 1 and stats[0, 0] == 0 and stats[0, 1] == 0      # total_error == 0 -> inactive (:584-586)
    pr, wrote = N.priorities(mn)
    assert wrote and (pr == 0).all()


def test_block_visit_rule():
    n = 4000
    assert all(sro.noise_block_fires(1.0, 7, 3, b, n) for b in range(200))
    assert not any(sro.noise_block_fires(0.0, 7, 3, b, n) for b in range(200))
    hits = sum(sro.noise_block_fires(0.25, 7, 3, b, n) for b in range(n))
    assert abs(hits / n - 0.25) < 0.03                   # Bernoulli(frac(priority)) (:196-197)
    a = [sro.noise_block_fires(0.5, 7, 3, b, n) for b in range(64)]
    assert a == [sro.noise_block_fires(0.5, 7, 3, b, n) for b in range(64)]
    assert a != [sro.noise_block_fires(0.5, 7, 4, b, n) for b in range(64)]


def test_golden_region_list():
    with open(os.path.join(HERE, "golden", "noise_golden.json")) as fh:
        G = json.load(fh)
    acc, aux = framebuffers(G["W"], G["H"], seed=G["fb_seed"])
    err, mn = sro.noise_error_map(acc, aux)
    N = sro.OracleNoise(G["W"], G["H"], seed=G["seed"])
    for _ in range(G["analyses"]):
        N.analyze(err, mn)
    rects, stats = N.regions()
    assert rects.tolist() == G["rects"]
    assert stats[:, 0].astype(int).tolist() == G["active"]
    assert np.allclose(stats[:, 1], np.array(G["noise"], np.float32), rtol=1e-6)
    pr, _ = N.priorities(mn)
    assert np.allclose(pr[:: G["priority_stride"]], np.array(G["priority_sample"], np.float32), rtol=1e-5, atol=1e-9)
