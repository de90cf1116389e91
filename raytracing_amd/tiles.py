"""Tile sharding of the frame across GPUs and the two exchange steps of the multi-GPU design.

The reference divides work over threads by interleaved pixel blocks with the immutable scene shared
(`SceneManager::assign_iterative_groups`, SceneManager.h:441-459; blocks at :944-950). Across GPUs the scene
is replicated and each rank renders a horizontal slab of tile rows; nothing is exchanged on the data path.
Two collectives exist, both outside the per-sample loop:

* `gather_framebuffer` — the reference's per-block copy into one output (`get_screen_update`,
  SceneManager.h:801-815) becomes an all-gather of each rank's slab over RCCL/xGMI (backend "nccl"), or gloo on CPU.
* `allreduce_noise_stats` — the per-region error sums and the minimum sample count of `analyze_noise`
  (SceneManager.h:488-502, 559-579) become all-reduce(sum) and all-reduce(min) of a few KB.

One process per GPU; works with world_size 1 without an initialised process group.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

TILE = 4   # the reference's block_size (SceneManager.h:299): slabs are cut at multiples of it


def slab(rank: int, world: int, height: int, tile: int = TILE):
    """Rows [y0, y1) of `rank`: tile rows dealt out as evenly as possible, contiguous per rank."""
    n_tiles = (height + tile - 1) // tile
    t0 = (n_tiles * rank) // world
    t1 = (n_tiles * (rank + 1)) // world
    return min(t0 * tile, height), min(t1 * tile, height)


def slabs(world: int, height: int, tile: int = TILE):
    return [slab(r, world, height, tile) for r in range(world)]


def _is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def gather_framebuffer(buf: torch.Tensor, height: int, width: int, tile: int = TILE):
    """In-place all-gather of a (height*width, C) framebuffer whose rank-local slab rows are valid."""
    if not _is_dist():
        return buf
    world, rank = dist.get_world_size(), dist.get_rank()
    view = buf.view(height, -1)
    cuts = slabs(world, height, tile)
    sizes = {y1 - y0 for y0, y1 in cuts}
    if len(sizes) == 1:
        y0, y1 = cuts[rank]
        dist.all_gather_into_tensor(view, view[y0:y1].contiguous())
    else:
        # ragged slabs: one broadcast per owner (each slab travels once per link)
        for r, (y0, y1) in enumerate(cuts):
            if y1 > y0:
                part = view[y0:y1].contiguous()
                dist.broadcast(part, src=r)
                view[y0:y1] = part
    return buf


def allreduce_noise_stats(sums: torch.Tensor, min_samples: torch.Tensor):
    """Combine adaptive-sampling statistics: `sums` (any shape, summed) and `min_samples` (minimum)."""
    if _is_dist():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(min_samples, op=dist.ReduceOp.MIN)
    return sums, min_samples


def luminance(rgb: torch.Tensor):
    """`ImageProcessing::luminance` weights (ImageProcessing.h:115-118)."""
    return 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]


def slab_noise_stats(accum4: torch.Tensor, aux4: torch.Tensor, height: int, width: int, y0: int, y1: int):
    """Per-pixel error |L - L_aux| / max(L^1.1, L) (SceneManager.h:566-579) summed over this rank's slab,
    plus the pixel count and the minimum sample count. Returns (sums[2], min_samples[1])."""
    a = accum4.view(height, width, 4)[y0:y1]
    x = aux4.view(height, width, 4)[y0:y1]
    L, La = luminance(a[..., :3]), luminance(x[..., :3])
    denom = torch.maximum(L.clamp_min(0) ** 1.1, L).clamp_min(1e-12)
    err = (L - La).abs() / denom
    sums = torch.stack([err.sum(), torch.tensor(float(err.numel()), device=err.device, dtype=err.dtype)])
    mins = a[..., 3].min().reshape(1) if a.numel() else torch.full((1,), float("inf"), device=a.device)
    return sums, mins
