"""BASELINE.json configs[1] at its full size on the GPU (262 144-triangle atrium, 1920x1080, max_generations 8): what the
bench renders, checked through size-independent properties plus oracle comparisons on strided subsets and a crop
(the CPU oracle needs ~10 s for one whole pass; the subsets take about a second)."""
import numpy as np
import pytest

from test_cast_gpu import _assert_hits_equal
from test_shade_gpu import _compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(tables):
    from oracle import sro
    from raytracing_amd import engine, scenes
    desc = scenes.atrium(262144, res=(1920, 1080), subdiv=8, max_generations=8)
    PS = engine.package(desc, tables=tables)
    RE = engine.RayEngine(0)
    RE.load_scene(PS)
    S = sro.OracleScene(desc, tables=tables)
    yield desc, PS, RE, S
    RE.close()


def test_package_and_every_97th_primary_ray(full):
    from oracle import sro
    desc, PS, RE, S = full
    on, op, ot = S.packaged()
    assert np.array_equal(on, PS.avx_bvh) and np.array_equal(op, PS.avx_tri_data) and np.array_equal(ot, PS.tri_data)
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    dirs, keys = sro.camera_rays_keyed(cam, (0, 0, w, h), 3, 20241111)
    org = np.tile(np.asarray(cam["position"], np.float32), (len(dirs), 1))
    g, gc = RE.execute_ray_cast(org, dirs)                      # all 2 073 600 primaries on the GPU
    assert (g["material"] != 0xFFFFFFFF).mean() > 0.9            # a closed atrium: nearly every pixel hits
    sel = np.arange(0, len(dirs), 97)
    c, cc = S.cast_closest(org[sel], dirs[sel])
    _assert_hits_equal(g[sel], c)
    g2, gc2 = RE.execute_ray_cast(org[sel], dirs[sel])
    assert int(gc2[0]) == int(cc[0]) and int(gc2[1]) == int(cc[1]) - int(cc[3])
    rg, _ = RE.process_ray(org[sel][::8], dirs[sel][::8], keys[sel][::8], generation=0, ray_capacity=1 << 22)
    rc, _ = S.process_rays_keyed(org[sel][::8], dirs[sel][::8], keys[sel][::8], generation=0)
    _compare(rg, rc)


def test_full_frame_pass_properties(full):
    import torch
    desc, PS, RE, S = full
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    aux = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    c0 = RE.trace_pass(cam, (0, 0, w, h), 0, 20241111, acc, aux, ray_capacity=1 << 26)
    c1 = RE.trace_pass(cam, (0, 0, w, h), 1, 20241111, acc, aux, ray_capacity=1 << 26)
    torch.cuda.synchronize()
    for c in (c0, c1):
        assert int(c["rays_processed"]) == w * h + int(c["diffuses_cast"])      # every queued child is processed once
        assert int(c["hits"]) <= int(c["rays_processed"]) and int(c["shadows_cast"]) == int(c["hits"]) * len(desc.suns)
        assert 25 * w * h < int(c["rays_processed"]) < 40 * w * h               # fan-out 32, 8, 2, 1... with Russian roulette
    n = acc[:, 3]
    assert float(n.max()) == 2.0 and float((n == 2).float().mean()) > 0.999     # a non-finite sample is skipped, rarely
    assert float(aux[:, 3].max()) == 1.0                                         # the aux buffer took pass 1 only
    assert bool(torch.isfinite(acc).all()) and float(acc[:, :3].mean()) > 0
    # the same two passes on a 64x36 crop by the oracle (keyed streams are per pixel and pass: a crop is the same samples)
    x0, y0, cw, ch = 928, 522, 64, 36
    acc_c = np.zeros((h * w, 4), np.float32); aux_c = np.zeros((h * w, 4), np.float32)
    for p in range(2):
        S.render_pass(cam, (x0, y0, cw, ch), p, 20241111, acc_c, aux_c, mode=0, threads=8)
    rows = (np.arange(y0, y0 + ch)[:, None] * w + np.arange(x0, x0 + cw)[None, :]).reshape(-1)
    a = acc.cpu().numpy()
    assert np.array_equal(a[rows, 3], acc_c[rows, 3])
    _compare(a[rows, :3], acc_c[rows, :3])
    # one call with both passes = the two calls above
    acc2 = torch.zeros_like(acc); aux2 = torch.zeros_like(aux)
    c2 = RE.trace_pass(cam, (0, 0, w, h), 0, 20241111, acc2, aux2, ray_capacity=1 << 26, n_passes=2)
    torch.cuda.synchronize()
    assert int(c2["rays_processed"]) == int(c0["rays_processed"]) + int(c1["rays_processed"])
    assert int(c2["nodes"]) == int(c0["nodes"]) + int(c1["nodes"]) and int(c2["packets"]) == int(c0["packets"]) + int(c1["packets"])
    assert bool(torch.equal(acc2[:, 3], acc[:, 3]))
    assert bool(torch.allclose(acc2[:, :3], acc[:, :3], rtol=1e-4, atol=1e-6))
