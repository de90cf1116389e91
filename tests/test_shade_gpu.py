"""GPU parity of the shade/bounce path against the CPU oracle run with the same keyed RNG.

SURVEY §8 rows a9-a18: srx_camera_rays / srx_process_rays / srx_trace_pass vs the restated
Camera::random_slope_at (Camera.h:138), RayEngine::process_ray (Core.h:359-601) and the accumulation of
RenderThread::process_iteratively (SceneManager.h:220-233). Both sides draw identical random numbers per
ray, so a sample differs only by libm/pow rounding (<=1e-4 relative) unless a comparison against a random
number flips; flipped samples are counted and bounded, the image mean must agree to 1e-3.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 2e-4          # per-sample relative tolerance (fp32 shading, double-promoted terms via pow -> multiplies)
FLIP_FRACTION = 0.02    # samples allowed to differ more than REL_TOL (a ξ<p decision flipped by 1 ulp)
MEAN_TOL = 2e-3         # relative tolerance on the mean radiance of the whole batch


def _scenes():
    from raytracing_amd import scenes
    return {
        "cornell": lambda: scenes.cornell(64, 2, with_sun=True),
        "atrium": lambda: scenes.atrium(16384, res=(96, 54), subdiv=2, tex_size=64),
        "interior": lambda: scenes.interior(8000, res=(80, 45), subdiv=2, tex_size=64),
        # 50x50 textures: not a multiple of the 32x32 Morton tile, exercises the padded tiles
        # C5 (BASELINE.json configs[4]): closed diffuse box + clutter, 16 generations, monte_carlo_max 1, no sun
        "diffuse_box": lambda: scenes.diffuse_box(3000, res=(64, 36), subdiv=2),
        "interior_npot": lambda: scenes.interior(6000, res=(64, 36), subdiv=2, tex_size=50),
        # C5 "and also 32" (Scene.h:29): the same box with the reference's default fan-out 32 / 8 / 2, then 13 generations of 1
        "diffuse_box_mcm32": lambda: scenes.diffuse_box(3000, res=(48, 27), subdiv=2, monte_carlo_max=32),
    }


@pytest.fixture(scope="module", params=["cornell", "atrium", "interior", "diffuse_box", "interior_npot", "diffuse_box_mcm32"])
def loaded(request, tables):
    from oracle import sro
    from raytracing_amd import engine
    desc = _scenes()[request.param]()
    PS = engine.package(desc, tables=tables)
    RE = engine.RayEngine(0)
    RE.load_scene(PS)
    S = sro.OracleScene(desc, tables=tables)
    yield request.param, desc, RE, S
    RE.close()


def test_camera_rays_bit_exact(loaded):
    from oracle import sro
    from raytracing_amd import engine
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    for p in (0, 3):
        dg, kg = engine.camera_rays(cam, (0, 0, w, h), p, 77)
        dc, kc = sro.camera_rays_keyed(cam, (0, 0, w, h), p, 77)
        assert np.array_equal(kg, kc)
        assert np.array_equal(dg.view(np.uint32), dc.view(np.uint32))


def _compare(rg, rc):
    assert np.isfinite(rc).all() == np.isfinite(rg).all()
    fin = np.isfinite(rc).all(1) & np.isfinite(rg).all(1)
    err = np.abs(rg[fin] - rc[fin]).max(1)
    scale = np.maximum(np.abs(rc[fin]).max(1), 1e-3)
    bad = (err / scale) > REL_TOL
    assert bad.mean() <= FLIP_FRACTION, f"{bad.mean():.4f} of samples differ"
    mg, mc = rg[fin].mean(0), rc[fin].mean(0)
    assert np.allclose(mg, mc, rtol=MEAN_TOL, atol=1e-5), (mg, mc)
    return bad.mean()


def test_process_ray_matches_oracle(loaded):
    from oracle import sro
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    dirs, keys = sro.camera_rays_keyed(cam, (0, 0, w, h), 0, 4242)
    org = np.tile(np.asarray(cam["position"], np.float32), (len(dirs), 1))
    rg, cg = RE.process_ray(org, dirs, keys, generation=0)
    rc, st = S.process_rays_keyed(org, dirs, keys, generation=0)
    flips = _compare(rg, rc)
    # identical random numbers -> (nearly) identical ray trees
    assert abs(int(cg["rays_processed"]) - int(st[0])) <= 0.02 * int(st[0]) + 8
    assert abs(int(cg["shadows_cast"]) - int(st[2])) <= 0.02 * int(st[2]) + 8
    if flips == 0.0:
        assert int(cg["rays_processed"]) == int(st[0])
        assert int(cg["nodes"]) == int(st[3]) and int(cg["packets"]) == int(st[4]) - int(st[8])


def test_process_ray_deep_generation(loaded):
    """Rays injected at generation 3 (fan-out 1) and at max_generations (direct light only)."""
    from oracle import sro
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    dirs, keys = sro.camera_rays_keyed(cam, (0, 0, w, h // 2), 1, 9)
    org = np.tile(np.asarray(cam["position"], np.float32), (len(dirs), 1))
    for gen in (3, desc.max_generations):
        rg, cg = RE.process_ray(org, dirs, keys, generation=gen)
        rc, st = S.process_rays_keyed(org, dirs, keys, generation=gen)
        _compare(rg, rc)
        if gen == desc.max_generations:
            assert int(cg["diffuses_cast"]) == 0 and int(st[1]) == 0


def test_small_queue_gives_same_result(loaded):
    """A tiny ray_capacity forces many small batches and depth-first scheduling; results must not change
    beyond float-atomic summation order."""
    from oracle import sro
    from raytracing_amd import engine
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    dirs, keys = sro.camera_rays_keyed(cam, (0, 0, w, 8), 0, 31)
    org = np.tile(np.asarray(cam["position"], np.float32), (len(dirs), 1))
    r_big, c_big = RE.process_ray(org, dirs, keys, generation=0, ray_capacity=1 << 22)
    RE2 = engine.RayEngine(0)
    RE2.load_scene(RE.data)
    r_small, c_small = RE2.process_ray(org, dirs, keys, generation=0, ray_capacity=4096)
    RE2.close()
    assert int(c_big["rays_processed"]) == int(c_small["rays_processed"])
    if desc.monte_carlo_max > 1:      # fan-out 32/8/2: 4 096 slots cannot hold a generation (the fan-out-1 box fits)
        assert int(c_small["trace_launches"]) > int(c_big["trace_launches"])
    assert np.allclose(r_big, r_small, rtol=1e-4, atol=1e-6, equal_nan=True)


def test_trace_pass_accumulation(loaded):
    import torch
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    aux = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    acc_c = np.zeros((h * w, 4), np.float32)
    aux_c = np.zeros((h * w, 4), np.float32)
    for p in range(3):
        torch.cuda.synchronize()
        cg = RE.trace_pass(cam, (0, 0, w, h), p, 1234, acc, aux)
        S.render_pass(cam, (0, 0, w, h), p, 1234, acc_c, aux_c, mode=0, threads=8)
    torch.cuda.synchronize()
    a, x = acc.cpu().numpy(), aux.cpu().numpy()
    assert np.array_equal(a[:, 3], acc_c[:, 3]) or np.abs(a[:, 3] - acc_c[:, 3]).sum() <= 4   # sample counts
    assert (x[:, 3] <= 1).all() and x[:, 3].sum() > 0                                        # aux took pass 1 only
    _compare(a[:, :3], acc_c[:, :3])
    _compare(x[:, :3], aux_c[:, :3])
    assert int(cg["rays_processed"]) > w * h


def test_interleaved_parts_compose_to_the_full_pass(loaded):
    """srx_trace_pass_interleaved: the shares of 3 callers (tile rows dealt round-robin, a short last group) touch
    disjoint pixels and together equal the single full-frame pass (same keyed streams per pixel; float atomics may
    reorder a pixel's additions, hence the small tolerance)."""
    import torch
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    full = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    parts = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    c_full = RE.trace_pass(cam, (0, 0, w, h), 2, 99, full)
    rays = 0
    for part in range(3):
        before = parts.clone()
        c = RE.trace_pass(cam, (0, 0, w, h), 2, 99, parts, interleave=(4, part, 3))
        rays += int(c["rays_processed"])
        touched = (parts[:, 3] != before[:, 3]).view(h, w).any(dim=1).cpu().numpy()
        expect = np.array([(y // 4) % 3 == part for y in range(h)])
        assert np.array_equal(touched, expect)
    torch.cuda.synchronize()
    assert rays == int(c_full["rays_processed"])
    a, b = full.cpu().numpy(), parts.cpu().numpy()
    assert np.array_equal(a[:, 3], b[:, 3])
    assert np.allclose(a[:, :3], b[:, :3], rtol=1e-4, atol=1e-6)


def test_several_passes_in_one_call_equal_one_call_per_pass(loaded):
    """srx_trace_passes_interleaved: 3 passes in one call (their rays share every launch) = 3 calls of one pass: same
    sample counts in both buffers, same rays and traversal counters, radiance equal up to float-atomic summation
    order; a share of interleaved rows behaves the same; too many passes for the 24-bit sample slots fail loudly."""
    import torch
    from raytracing_amd import _lib
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    for interleave in (None, (4, 1, 2)):
        one = [torch.zeros((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        many = [torch.zeros((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        tot = {}
        for p in range(1, 4):
            c = RE.trace_pass(cam, (0, 0, w, h), p, 77, one[0], one[1], interleave=interleave)
            for k in ("rays_processed", "shadows_cast", "nodes", "packets", "hits"):
                tot[k] = tot.get(k, 0) + int(c[k])
        c3 = RE.trace_pass(cam, (0, 0, w, h), 1, 77, many[0], many[1], interleave=interleave, n_passes=3)
        torch.cuda.synchronize()
        for k, v in tot.items():
            assert int(c3[k]) == v, k
        assert int(c3["trace_launches"]) < 3 * int(c["trace_launches"])
        for a, b in zip(one, many):
            a, b = a.cpu().numpy(), b.cpu().numpy()
            assert np.array_equal(a[:, 3], b[:, 3])
            assert np.allclose(a[:, :3], b[:, :3], rtol=1e-4, atol=1e-6, equal_nan=True)
        assert float(many[1][:, 3].max()) == 2.0            # aux took passes 1 and 3
    with pytest.raises(_lib.SrxError):
        RE.trace_pass(cam, (0, 0, w, h), 0, 77, many[0], many[1], n_passes=(1 << 24) // (w * h) + 1)
    with pytest.raises(_lib.SrxError):
        RE.trace_pass(cam, (0, 0, w, h), 0, 77, many[0], many[1], n_passes=0)


def test_shadow_stream_on_and_off_give_the_same_pass(loaded, monkeypatch):
    """The shadow kernels run on a second stream beside the next generation's closest-hit kernel (two shadow queues,
    event-ordered); SRX_SHADOW_STREAM=0 puts them back on the main stream. Same rays, same counters, same image up
    to the order of the float atomics."""
    import torch
    from raytracing_amd import engine
    name, desc, RE, S = loaded
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SRX_SHADOW_STREAM", flag)
        RE2 = engine.RayEngine(0)
        RE2.load_scene(RE.data)
        acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
        aux = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
        cs = [RE2.trace_pass(cam, (0, 0, w, h), p, 31, acc, aux, ray_capacity=1 << 16) for p in range(2)]   # small queue: many batches
        torch.cuda.synchronize()
        out.append((acc.cpu().numpy(), aux.cpu().numpy(), cs))
        RE2.close()
    (a1, x1, c1), (a0, x0, c0) = out
    for ca, cb in zip(c1, c0):
        for k in ("rays_processed", "shadows_cast", "nodes", "packets", "shadow_nodes", "shadow_packets", "hits"):
            assert int(ca[k]) == int(cb[k]), k
    assert np.array_equal(a1[:, 3], a0[:, 3]) and np.array_equal(x1[:, 3], x0[:, 3])
    assert np.allclose(a1[:, :3], a0[:, :3], rtol=1e-4, atol=1e-6, equal_nan=True)
    assert np.allclose(x1[:, :3], x0[:, :3], rtol=1e-4, atol=1e-6, equal_nan=True)


def test_uniform_direction_shadow_form_equals_the_general_form(loaded, monkeypatch):
    """A scene with one sun sends every shadow ray of the renderer in the same direction (SunLight::vectorTo,
    Primitives.h:39-51): k_trace<3> keeps that direction, its reciprocal, the leaf filter's per-axis widening factors and
    the guard's quantised direction in scalar registers. SRX_SHADOW_UNIFORM=0 runs the general form (k_trace<2>, per-ray
    values from LDS). Same shadow answers: same counters, node and packet tests included, same image."""
    import torch
    from raytracing_amd import engine
    name, desc, RE, S = loaded
    if len(desc.suns) != 1:
        pytest.skip("the uniform form is for scenes with exactly one sun")
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SRX_SHADOW_UNIFORM", flag)
        acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
        cs = [RE.trace_pass(cam, (0, 0, w, h), p, 17, acc, None) for p in range(2)]
        torch.cuda.synchronize()
        out.append((acc.cpu().numpy(), cs))
    (a1, c1), (a0, c0) = out
    for ca, cb in zip(c1, c0):
        for k in ("rays_processed", "shadows_cast", "nodes", "packets", "shadow_nodes", "shadow_packets", "hits"):
            assert int(ca[k]) == int(cb[k]), k
        assert int(ca["shadows_cast"]) > 0
    assert np.array_equal(a1[:, 3], a0[:, 3])
    assert np.allclose(a1[:, :3], a0[:, :3], rtol=1e-4, atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("mcm,cap", [(1, 1 << 12), (1, 1 << 11), (32, 1 << 15)])
def test_chains_back_to_back(tables, monkeypatch, mcm, cap):
    """Chain mode (generations of fan-out 1 enqueued back to back, counts read on the device, csrc/srx_render.hip
    process_chain) against SRX_CHAIN=0 (a host round trip per generation) where SEVERAL chains run in one call: a small
    ray_capacity splits a segment into batches, and with fan-out 32/8/2 the depth-first recursion starts a chain per
    batch of generation 3. Scene with a sun: the shadow kernels of one chain's last links (second stream, lowest
    priority) are still dispatching blocks when the next chain begins; clearing the count ring under them dropped their
    shadow rays. Rays, shadow rays, every traversal counter and the image must equal the unchained run."""
    import torch
    from raytracing_amd import engine, scenes
    desc = scenes.atrium(16384, res=(96, 54), subdiv=2, tex_size=64, monte_carlo_max=mcm)
    PS = engine.package(desc, tables=tables)
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SRX_CHAIN", flag)
        RE = engine.RayEngine(0)
        RE.load_scene(PS)
        acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
        cs = [RE.trace_pass(cam, (0, 0, w, h), p, 5, acc, None, ray_capacity=cap) for p in range(3)]
        torch.cuda.synchronize()
        out.append((acc.cpu().numpy(), cs))
        RE.close()
    (a1, c1), (a0, c0) = out
    for ca, cb in zip(c1, c0):
        for k in ("rays_processed", "shadows_cast", "nodes", "packets", "shadow_nodes", "shadow_packets", "hits"):
            assert int(ca[k]) == int(cb[k]), k
        assert int(ca["shadows_cast"]) > 0
    assert int(c1[0]["trace_launches"]) > desc.max_generations + 1      # more launches than one chain has links: several chains per call
    assert np.array_equal(a1[:, 3], a0[:, 3])
    assert np.allclose(a1[:, :3], a0[:, :3], rtol=1e-4, atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("scene", ["atrium", "instanced"])
def test_shade_rounds_equal_one_launch(tables, monkeypatch, scene):
    """Rounds of the shade kernel (csrc/srx_render.hip launch_batch, csrc/srx_shade.hip: a wave whose bounce loop is down to a
    few busy lanes parks them as GCont records, a second launch packs the parked loops of the batch 64 to a wave and goes on
    where they stopped) against SRX_SHADE_ROUNDS=1 (every wave runs until its slowest ray is done, Core.h:498-591 as one
    loop). A parked loop resumes with its slots, attempts, carried throughput (quirk Q8) and random stream: the same rays, the same
    children and shadow rays, the same hits, texel fetches and PDF lookups, the same image. Also with three rounds and
    with thresholds that park nearly everything / nearly nothing."""
    import torch
    from raytracing_amd import engine, scenes
    if scene == "atrium":
        desc = scenes.atrium(16384, res=(96, 54), subdiv=2, tex_size=64, monte_carlo_max=32)
    else:
        desc = scenes.instanced(40, 200, res=(96, 54), subdiv=2, monte_carlo_max=32)      # open scene: misses, silhouettes with many rejected attempts
    PS = engine.package(desc, tables=tables)
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    out = []
    for env in ({"SRX_SHADE_ROUNDS": "1"}, {"SRX_SHADE_ROUNDS": "2"}, {"SRX_SHADE_ROUNDS": "3", "SRX_SHADE_SPILL_BELOW": "40"},
                {"SRX_SHADE_ROUNDS": "2", "SRX_SHADE_SPILL_BELOW": "63", "SRX_SHADE_PARK_PCT": "100"},
                {"SRX_SHADE_ROUNDS": "2", "SRX_SHADE_SPILL_BELOW": "1", "SRX_SHADE_PARK_PCT": "400"}):
        for k in ("SRX_SHADE_ROUNDS", "SRX_SHADE_SPILL_BELOW", "SRX_SHADE_PARK_PCT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        RE = engine.RayEngine(0)
        RE.load_scene(PS)
        acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
        cs = [RE.trace_pass(cam, (0, 0, w, h), p, 23, acc, None) for p in range(2)]
        torch.cuda.synchronize()
        out.append((acc.cpu().numpy(), cs))
        RE.close()
    a0, c0 = out[0]
    assert int(c0[0]["diffuses_cast"]) > 0 and int(c0[0]["pdf_lookups"]) > int(c0[0]["diffuses_cast"])
    for a, cs in out[1:]:
        for ca, cb in zip(cs, c0):
            for k in ("rays_processed", "diffuses_cast", "shadows_cast", "nodes", "packets", "shadow_nodes", "shadow_packets", "hits",
                      "texels", "pdf_lookups"):
                assert int(ca[k]) == int(cb[k]), k
            assert int(ca["shade_launches"]) > int(cb["shade_launches"])                  # the later rounds were launched
        assert np.array_equal(a[:, 3], a0[:, 3])
        assert np.allclose(a[:, :3], a0[:, :3], rtol=1e-4, atol=1e-6, equal_nan=True)


def test_converged_image_matches_reference_rng_estimator(tables):
    """Property test at image level: the GPU (keyed per-ray streams, throughput carried down the path, float atomics)
    and the oracle in REFERENCE mode (one sequential xoshiro stream, recursion multiplying on the way back,
    SceneManager.h:193-237) are two samplings of the same estimator. After 64 spp their images must agree to the
    Monte-Carlo noise: means within 1.5 %, 6x6-pixel tile means within 8 % RMS of the image mean."""
    import torch
    from oracle import sro
    from raytracing_amd import engine, scenes
    desc = scenes.cornell(48, 2, with_sun=True)
    PS = engine.package(desc, tables=tables)
    RE = engine.RayEngine(0)
    RE.load_scene(PS)
    S = sro.OracleScene(desc, tables=tables)
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    acc_c = np.zeros((h * w, 4), np.float32)
    for p in range(64):
        RE.trace_pass(cam, (0, 0, w, h), p, 2024, acc, None)
        S.render_pass(cam, (0, 0, w, h), p, 77, acc_c, None, mode=1, threads=8)
    torch.cuda.synchronize()
    a = acc.cpu().numpy()[:, :3].reshape(h, w, 3)
    c = acc_c[:, :3].reshape(h, w, 3)
    assert (acc.cpu().numpy()[:, 3] >= 62).all() and (acc_c[:, 3] == 64).all()
    assert np.allclose(a.mean((0, 1)), c.mean((0, 1)), rtol=1.5e-2)
    ta = a.reshape(h // 6, 6, w // 6, 6, 3).mean((1, 3))
    tc = c.reshape(h // 6, 6, w // 6, 6, 3).mean((1, 3))
    rms = np.sqrt(((ta - tc) ** 2).mean()) / c.mean()
    assert rms < 0.08, rms       # two CPU runs with different streams differ by 0.048
    RE.close()


def _psnr(a, b, peak=1.0):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())
    return 10.0 * np.log10(peak * peak / max(mse, 1e-300)), float(np.sqrt(mse))


def test_converged_image_psnr_against_the_cpu_path(tables):
    """BASELINE.json's image criterion, stated: after 256 spp of the Cornell scene (48x48, sun on), display-referred
    (clamp to [0, 1], gamma 2.2, the reference's canvas conversion with the view transform off; peak = 1):

    * GPU vs the CPU oracle drawing the SAME per-ray streams: PSNR >= 120 dB (RMSE <= 1e-6) — only the order of the float
      additions differs (measured 167 dB);
    * GPU vs the CPU oracle in REFERENCE mode (sequential xoshiro stream per thread, recursive estimator,
      SceneManager.h:193-237), i.e. independent samples of the same estimator: per pixel PSNR >= 29.5 dB (RMSE <= 3.4e-2),
      over 4x4-pixel tile means PSNR >= 38 dB, and neither more than 1.5 dB (tiles: 2 dB) below the PSNR between two CPU
      reference-mode runs with different seeds — the Monte-Carlo noise floor at this sample count (measured 30.9 vs
      30.9 dB per pixel, 41.9 vs 42.9 dB over tiles: the GPU image is as close to a CPU image as a second CPU image is)."""
    import torch
    from oracle import sro
    from raytracing_amd import engine, scenes
    desc = scenes.cornell(48, 2, with_sun=True)
    PS = engine.package(desc, tables=tables)
    RE = engine.RayEngine(0)
    RE.load_scene(PS)
    S = sro.OracleScene(desc, tables=tables)
    cam = desc.camera
    w, h = int(cam["res_x"]), int(cam["res_y"])
    spp = 256
    acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    same = np.zeros((h * w, 4), np.float32)
    ref_a = np.zeros((h * w, 4), np.float32)
    ref_b = np.zeros((h * w, 4), np.float32)
    for p in range(spp):
        RE.trace_pass(cam, (0, 0, w, h), p, 2024, acc, None)
        S.render_pass(cam, (0, 0, w, h), p, 2024, same, None, mode=0, threads=8)
        S.render_pass(cam, (0, 0, w, h), p, 77, ref_a, None, mode=1, threads=8)
        S.render_pass(cam, (0, 0, w, h), p, 991, ref_b, None, mode=1, threads=8)
    torch.cuda.synchronize()
    display = lambda x: np.clip(x[:, :3], 0.0, 1.0) ** (1.0 / 2.2)
    tiles4 = lambda x: x.reshape(h // 4, 4, w // 4, 4, 3).mean((1, 3))
    g, c_same, c_a, c_b = display(acc.cpu().numpy()), display(same), display(ref_a), display(ref_b)
    p_same, r_same = _psnr(g, c_same)
    p_ref, r_ref = _psnr(g, c_a)
    p_floor, r_floor = _psnr(c_a, c_b)
    t_ref, _ = _psnr(tiles4(g), tiles4(c_a))
    t_floor, _ = _psnr(tiles4(c_a), tiles4(c_b))
    print(f"PSNR same streams {p_same:.1f} dB (rmse {r_same:.2e}); vs reference mode {p_ref:.1f} dB (rmse {r_ref:.2e}), tiles {t_ref:.1f} dB; "
          f"CPU seed A vs B {p_floor:.1f} dB (rmse {r_floor:.2e}), tiles {t_floor:.1f} dB")
    assert p_same >= 120.0 and r_same <= 1e-6
    assert p_ref >= 29.5 and r_ref <= 3.4e-2
    assert p_ref >= p_floor - 1.5
    assert t_ref >= 38.0 and t_ref >= t_floor - 2.0
    RE.close()
