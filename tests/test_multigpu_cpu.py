"""world_size-2 gloo test (CPU) of the multi-GPU exchange steps: framebuffer all-gather of tile slabs and the
noise-statistics all-reduce. The data path itself has no collective."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import tiles
    y0, y1 = tiles.slab(rank, world, H)
    full = torch.arange(H * W * 4, dtype=torch.float32).view(H * W, 4)
    buf = torch.zeros_like(full)
    buf.view(H, W * 4)[y0:y1] = full.view(H, W * 4)[y0:y1]
    tiles.gather_framebuffer(buf, H, W)
    ok_gather = bool(torch.equal(buf, full))
    acc = torch.rand(H * W, 4, generator=torch.Generator().manual_seed(1)) + 0.1
    aux = torch.rand(H * W, 4, generator=torch.Generator().manual_seed(2)) + 0.1
    acc[:, 3] = 5 + rank
    sums, mins = tiles.slab_noise_stats(acc, aux, H, W, y0, y1)
    tiles.allreduce_noise_stats(sums, mins)
    ref_sums, _ = tiles.slab_noise_stats(acc, aux, H, W, 0, H)
    q.put((rank, ok_gather, sums.tolist(), mins.tolist(), ref_sums.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [16, 18])      # 18 rows -> ragged slabs (8 + 10)
def test_gather_and_allreduce_world2(H):
    W, world, port = 6, 2, 29500 + os.getpid() % 2000 + H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sums, mins, ref in res:
        assert ok
        assert np.allclose(sums, ref, rtol=1e-5)
        assert mins == [5.0]
