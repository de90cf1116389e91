"""Noise-weighted dealing of the tile rows to the ranks (srx_deal, csrc/srx_deal.hip): the counterpart of the
second half of SceneManager::balance_threads (SceneManager.h:693-773). Host code, so all of it runs without a GPU: the
tables, the rebalance (deterministic, a function of the block priorities alone, burdens within the reference's 5 %),
and — world 2 over gloo, the CPU oracle rendering each rank's rows — that frames do not depend on the deal."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _check_tables(d):
    owner, slot, _ = d.tables()
    assert len(owner) == d.n_groups and owner.min() >= 0 and owner.max() < d.nparts
    seen = []
    for p in range(d.nparts):
        g = d.groups_of(p)
        assert sorted(slot[g].tolist()) == list(range(len(g)))          # a part's slots are 0 .. n - 1
        seen += g.tolist()
        short = d.res_y % d.tile and (d.n_groups - 1) in g.tolist()
        if short:
            assert g[-1] == d.n_groups - 1                                # a short last group is last in its owner's list
    assert sorted(seen) == list(range(d.n_groups))                      # every group exactly once
    rows = torch.cat([d.rows_of(p) for p in range(d.nparts)])
    assert sorted(rows.tolist()) == list(range(d.res_y))


def test_a_deal_starts_as_the_round_robin():
    from raytracing_amd import tiles
    for H, n in ((64, 2), (54, 3), (1080, 8), (22, 4)):
        d = tiles.Deal(40, H, n)
        _check_tables(d)
        for p in range(n):
            assert torch.equal(d.rows_of(p), tiles.interleaved_rows(p, n, H))


def _reference_rebalance(lists, cost, max_rounds):
    """SceneManager.h:708-762 restated in Python on row groups, with the documented differences of srx_deal_rebalance
    (the first group from the front lighter than the gap moves, not the front one whatever it weighs; groups without
    burden stay; the extremes are taken afresh after every round, and rounds go on until one moves nothing) — float32
    arithmetic like the reference's."""
    f = np.float32
    T = len(lists)
    burden = [f(sum((cost[g] for g in L), f(0))) for L in lists]
    for p, L in enumerate(lists):
        b = f(0)
        for g in L:
            b = f(b + cost[g])
        burden[p] = b
    moved = 0
    for _ in range(max_rounds):
        before = moved
        mx, mn, mxi, mni = burden[0], burden[0], 0, 0
        for t in range(1, T):
            if burden[t] > mx: mx, mxi = burden[t], t
            if burden[t] < mn: mn, mni = burden[t], t
        with np.errstate(all="ignore"):
            sim = f(f(mx - mn) / mx)
        while sim > f(0.05) and mxi != mni and len(lists[mxi]) > 1:
            k = next((i for i, g in enumerate(lists[mxi]) if cost[g] > 0 and cost[g] < f(mx - mn)), None)
            if k is None:
                break
            g = lists[mxi].pop(k)
            lists[mni].append(g)
            mx, mn = f(mx - cost[g]), f(mn + cost[g])
            moved += 1
            with np.errstate(all="ignore"):
                sim = f(f(mx - mn) / mx)
        burden[mni], burden[mxi] = mn, mx
        mx, mn = max(burden), min(burden)
        with np.errstate(all="ignore"):
            sim = f(f(mx - mn) / mx)
        if not (sim > f(0.05) and moved > before):
            break
    return moved


@pytest.mark.parametrize("n", [2, 3, 8])
def test_rebalance_evens_the_burdens_and_is_deterministic(n):
    from raytracing_amd import tiles
    W, H = 256, 1080                                   # 270 tile rows of 64 blocks (a 1080p frame's rows)
    rng = np.random.default_rng(3)
    bx, by = W // 4, H // 4
    pr = (0.02 + 0.2 * rng.random((by, bx))).astype(np.float32)
    pr += (0.08 * (np.arange(by) % n))[:, None].astype(np.float32)   # cost grows with (tile row mod n): the round-robin's worst case
    d = tiles.Deal(W, H, n)
    owner0, _, _ = d.tables()
    cost = pr.sum(1, dtype=np.float64)
    b0 = np.array([cost[owner0 == p].sum() for p in range(n)])
    assert (b0.max() - b0.min()) / b0.max() > 0.3
    moved = d.rebalance(pr)
    _check_tables(d)
    owner, slot, burdens = d.tables()
    assert moved > 0 and (burdens.max() - burdens.min()) / burdens.max() <= 0.05       # the reference's 5 % (:729)
    assert np.allclose(burdens, [cost[owner == p].sum() for p in range(n)], rtol=1e-5)
    # a function of the priorities alone: a second deal of another "rank" arrives at the same tables, and nothing moves again
    d2 = tiles.Deal(W, H, n)
    assert d2.rebalance(torch.from_numpy(pr)) == moved
    assert all(np.array_equal(a, b) for a, b in zip(d.tables()[:2], d2.tables()[:2]))
    assert d.rebalance(pr) == 0
    # the same algorithm restated in Python gives the same lists
    lists = [list(range(p, by, n)) for p in range(n)]
    cost32 = np.zeros(by, np.float32)
    for g in range(by):
        c = np.float32(0)
        for x in range(bx):
            c = np.float32(c + pr[g, x])
        cost32[g] = c
    assert _reference_rebalance(lists, cost32, 64 * n) == moved
    for p in range(n):
        assert d.groups_of(p).tolist() == lists[p]


def test_rebalance_leaves_a_poisoned_controller_alone():
    """Quirk Q11: NaN priorities (SceneManager.h:571 on a negative luminance) make every burden NaN, every comparison of
    :711-729 false, and nothing moves, in the reference as here; all-zero priorities likewise."""
    from raytracing_amd import _lib, tiles
    d = tiles.Deal(64, 64, 2)
    before = d.tables()[0].copy()
    assert d.rebalance(np.full((16, 16), np.nan, np.float32)) == 0
    assert d.rebalance(np.zeros((16, 16), np.float32)) == 0
    assert np.array_equal(d.tables()[0], before)
    with pytest.raises(_lib.SrxError):
        d.rebalance(np.ones((3, 3), np.float32))                     # not this frame's blocks
    with pytest.raises(_lib.SrxError):
        tiles.Deal(64, 64, 2, tile=6).rebalance(np.ones((16, 16), np.float32))   # groups must be whole block rows


def _scene_and_priorities():
    """A tall 32 x 256 view of the Cornell box (64 tile rows) and priorities that alternate with the tile row, so that the
    round-robin gives one rank all the expensive rows."""
    from raytracing_amd import scenes
    desc = scenes.cornell(64, 1, with_sun=True)
    desc.monte_carlo_max = 2
    W, H = 32, 256
    cam = desc.camera.copy()
    cam["res_x"], cam["res_y"], cam["lens_width"], cam["lens_height"] = W, H, 0.5, 4.0
    pr = np.full((H // 4, W // 4), 0.3, np.float32)
    pr[1::2] = 0.7
    return desc, cam, W, H, pr


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import sro
    from raytracing_amd import tiles
    from test_deal_cpu import _scene_and_priorities
    desc, cam, W, H, pr = _scene_and_priorities()
    S = sro.OracleScene(desc, build_tables=True)
    deal = tiles.Deal(W, H, world)
    acc = np.zeros((H * W, 4), np.float32); aux = np.zeros((H * W, 4), np.float32)
    t_acc, t_aux = torch.from_numpy(acc), torch.from_numpy(aux)      # views of the same memory
    ones = np.ones(pr.size, np.float32)

    def visits(first, count, priority):
        for v in range(first, first + count):
            for g in deal.groups_of(rank):
                S.render_pass_adaptive(cam, (0, int(g) * tiles.TILE, W, tiles.TILE), v, 11, acc, aux, block_priority=priority, threads=1)

    visits(0, 2, ones)
    tiles.gather_framebuffer_dealt([t_acc, t_aux], deal)             # a rank that takes rows over needs their pixels: under the OLD deal
    moved = deal.rebalance(pr)
    _, _, burdens = deal.tables()
    visits(2, 3, pr.reshape(-1))
    tiles.gather_framebuffer_dealt([t_acc, t_aux], deal)
    q.put((rank, moved, burdens.tolist(), deal.tables()[0].tolist(), acc.copy(), aux.copy()))
    dist.destroy_process_group()


def test_frames_do_not_depend_on_the_deal_world2():
    """Two ranks over gloo, the CPU oracle rendering each rank's tile rows: round-robin visits, framebuffer gather, re-deal
    by the priorities, weighted visits, gather. The frames on both ranks must be IDENTICAL to one process rendering the
    whole frame with the same priorities (a pixel's samples are keyed by pixel and visit, not by who renders it), and the
    two ranks' expected work after the re-deal within 5 % of each other."""
    world, port = 2, 35500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import sro
    desc, cam, W, H, pr = _scene_and_priorities()
    S = sro.OracleScene(desc, build_tables=True)
    acc = np.zeros((H * W, 4), np.float32); aux = np.zeros((H * W, 4), np.float32)
    for v in range(5):
        S.render_pass_adaptive(cam, (0, 0, W, H), v, 11, acc, aux, block_priority=(np.ones(pr.size, np.float32) if v < 2 else pr.reshape(-1)), threads=4)
    assert res[0][1] == res[1][1] > 0 and res[0][3] == res[1][3]                 # both ranks made the same deal
    b = np.array(res[0][2])
    assert (b.max() - b.min()) / b.max() <= 0.05
    static = np.array([g % 2 for g in range(H // 4)])
    assert not np.array_equal(np.array(res[0][3]), static)                      # and it is not the static one
    for rank, _, _, _, a, x in res:
        assert np.array_equal(a.view(np.uint32), acc.view(np.uint32)), rank
        assert np.array_equal(x.view(np.uint32), aux.view(np.uint32)), rank
    assert acc[:, 3].max() == 5 and acc[:, 3].min() >= 2
