"""Tile sharding of the frame across GPUs and the two exchange steps of the multi-GPU design.

The reference divides work over threads by interleaved pixel blocks with the immutable scene shared
(`SceneManager::assign_iterative_groups`, SceneManager.h:441-459; blocks at :944-950). Across GPUs the scene
is replicated and each rank renders its share of the tile rows — dealt round-robin (`interleaved_rows`, the default,
which evens out cheap and expensive image regions) or as one contiguous slab (`slab`); nothing is exchanged on the
data path.
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


def interleaved_rows(rank: int, world: int, height: int, tile: int = TILE) -> torch.Tensor:
    """Image rows of `rank` when tile rows (groups of `tile` rows) are dealt round-robin: tile row t belongs to
    rank t % world. Matches srx_trace_pass_interleaved(group_rows=tile, part=rank, nparts=world) over the frame."""
    rows = torch.arange(height)
    return rows[(rows // tile) % world == rank]


def _is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_staged(fn, buf, *args):
    """gloo moves host memory only: with that backend a device framebuffer is exchanged through a host copy
    (CPU tests and single-GPU rehearsals; RCCL takes the device tensor as it is)."""
    if buf.is_cuda and dist.get_backend() == "gloo":
        host = buf.cpu()
        fn(host, *args)
        buf.copy_(host)
        return buf
    return fn(buf, *args)


def gather_framebuffer(buf: torch.Tensor, height: int, width: int, tile: int = TILE):
    """In-place all-gather of a (height*width, C) framebuffer whose rank-local slab rows are valid."""
    if not _is_dist():
        return buf
    return _host_staged(_gather_slabs, buf, height, width, tile)


def _gather_slabs(buf, height, width, tile):
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


def gather_framebuffer_interleaved(buf, height: int, width: int, tile: int = TILE):
    """In-place all-gather of (height*width, C) framebuffers whose rank-owned interleaved rows are valid: `buf` is one
    tensor or a list of tensors of the same shape and dtype (main and aux buffer), which then travel in ONE collective.
    Each rank packs its rows, one all-gather moves the packed shares (padded to the largest), every rank unpacks with
    one indexed copy."""
    bufs = list(buf) if isinstance(buf, (list, tuple)) else [buf]
    if not _is_dist() or not bufs:
        return buf
    if bufs[0].is_cuda and dist.get_backend() == "gloo":      # gloo moves host memory only (CPU tests, rehearsals)
        hosts = [b.cpu() for b in bufs]
        _gather_interleaved(hosts, height, width, tile)
        for b, h in zip(bufs, hosts):
            b.copy_(h)
    else:
        _gather_interleaved(bufs, height, width, tile)
    return buf


_LAYOUT = {}


def _interleaved_layout(world, height, tile, device):
    """Per layout, built once (the gather sits inside timed regions): this and every rank's row indices, and for the
    unpack the frame rows in the order the gathered buffer holds them with their positions in it (padding skipped)."""
    key = (world, height, tile, str(device))
    if key not in _LAYOUT:
        owned = [interleaved_rows(r, world, height, tile) for r in range(world)]
        most = max(len(o) for o in owned)
        dst = torch.cat(owned)
        src = torch.cat([r * most + torch.arange(len(o)) for r, o in enumerate(owned)])
        _LAYOUT[key] = ([o.to(device) for o in owned], most, dst.to(device), src.to(device))
    return _LAYOUT[key]


def _gather_interleaved(bufs, height, width, tile):
    world, rank = dist.get_world_size(), dist.get_rank()
    views = [b.view(height, -1) for b in bufs]
    cols = views[0].shape[1]
    owned, most, dst, src = _interleaved_layout(world, height, tile, bufs[0].device)
    nb = len(views)
    # [rank][buffer][row]: one rank's share of every buffer is contiguous
    mine = torch.zeros((nb, most, cols), dtype=bufs[0].dtype, device=bufs[0].device)
    for i, v in enumerate(views):
        mine[i, : len(owned[rank])] = v[owned[rank]]
    everyone = torch.empty((world, nb, most, cols), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(everyone.view(world * nb * most, cols), mine.view(nb * most, cols))
    for i, v in enumerate(views):
        v[dst] = everyone[:, i].reshape(world * most, cols)[src]
    return bufs


# ---------------------------------------------------------------- noise-weighted dealing of the tile rows (csrc/srx_deal.hip)
class Deal:
    """`srx_deal` of include/srx.h: which rank owns which tile row. Starts as the round-robin of `interleaved_rows`
    (`SceneManager::assign_iterative_groups`, SceneManager.h:441-459); `rebalance(block_priority)` re-deals the rows the way
    `SceneManager::balance_threads` (:693-773) re-deals blocks to threads, until the ranks' burdens (sums of block
    priorities) are within 5 %. A function of the priorities alone, so every rank computes the same deal; host code, works
    without a GPU."""

    def __init__(self, res_x, res_y, nparts, tile=TILE):
        import ctypes as C
        from . import _lib
        self.L, self.C, self._check = _lib.lib(), C, _lib.check
        self.h = C.c_void_p()
        self._check(self.L.srx_deal_create(C.c_int32(res_x), C.c_int32(res_y), C.c_int32(tile), C.c_int32(nparts), C.byref(self.h)), "srx_deal_create")
        self.res_x, self.res_y, self.nparts, self.tile = int(res_x), int(res_y), int(nparts), int(tile)
        self.n_groups = (self.res_y + tile - 1) // tile
        self._rows = {}

    def close(self):
        if self.h:
            self.L.srx_deal_destroy(self.h)
            self.h = self.C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rebalance(self, block_priority) -> int:
        """`block_priority`: the controller's priorities (torch tensor on any device, or an array), one per 4x4 block.
        Returns how many tile rows changed hands."""
        import numpy as np
        C = self.C
        pr = block_priority.detach().cpu().numpy() if hasattr(block_priority, "detach") else np.asarray(block_priority)
        pr = np.ascontiguousarray(pr, np.float32)
        moved = C.c_int32(0)
        self._check(self.L.srx_deal_rebalance(self.h, pr.ctypes.data_as(C.c_void_p), C.c_uint32(pr.size), C.byref(moved)), "srx_deal_rebalance")
        self._rows = {}
        return int(moved.value)

    def tables(self):
        """(owner[n_groups], slot[n_groups], burdens[nparts]) as numpy arrays."""
        import numpy as np
        C = self.C
        owner = np.zeros(self.n_groups, np.int32); slot = np.zeros(self.n_groups, np.int32); burdens = np.zeros(self.nparts, np.float32)
        n = C.c_uint32(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.L.srx_deal_get(self.h, p(owner), p(slot), C.c_uint32(self.n_groups), C.byref(n), p(burdens)), "srx_deal_get")
        return owner, slot, burdens

    def groups_of(self, part):
        """The tile rows of `part` in the order it renders and packs them."""
        import numpy as np
        owner, slot, _ = self.tables()
        g = np.nonzero(owner == part)[0]
        return g[np.argsort(slot[g])]

    def rows_of(self, part) -> torch.Tensor:
        """Image rows of `part`, in the order of its packed share."""
        if part not in self._rows:
            rows = []
            for g in self.groups_of(part):
                rows.extend(range(int(g) * self.tile, min(self.res_y, (int(g) + 1) * self.tile)))
            self._rows[part] = torch.tensor(rows, dtype=torch.long)
        return self._rows[part]


def gather_framebuffer_dealt(buf, deal: Deal):
    """`gather_framebuffer_interleaved` under a deal, over torch.distributed (gloo tests and rehearsals; on GPUs
    `Group.gather_framebuffer_dealt` does it through the library): every rank's rows, in its deal order, padded to the
    largest share, ONE all-gather for all the buffers, one indexed copy back."""
    bufs = list(buf) if isinstance(buf, (list, tuple)) else [buf]
    if not _is_dist() or not bufs:
        return buf
    staged = bufs[0].is_cuda and dist.get_backend() == "gloo"
    work = [b.cpu() for b in bufs] if staged else bufs
    world, rank = dist.get_world_size(), dist.get_rank()
    if world != deal.nparts:
        raise ValueError(f"the deal has {deal.nparts} parts, the job {world} ranks")
    H = deal.res_y
    views = [b.view(H, -1) for b in work]
    cols = views[0].shape[1]
    owned = [deal.rows_of(r).to(work[0].device) for r in range(world)]
    most = max(len(o) for o in owned)
    nb = len(views)
    mine = torch.zeros((nb, most, cols), dtype=work[0].dtype, device=work[0].device)
    for i, v in enumerate(views):
        mine[i, : len(owned[rank])] = v[owned[rank]]
    everyone = torch.empty((world, nb, most, cols), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(everyone.view(world * nb * most, cols), mine.view(nb * most, cols))
    for i, v in enumerate(views):
        for r in range(world):
            if r != rank:
                v[owned[r]] = everyone[r, i, : len(owned[r])]
    if staged:
        for b, h in zip(bufs, work):
            b.copy_(h)
    return buf


# ---------------------------------------------------------------- the C-ABI exchange (csrc/srx_group.hip)
class Group:
    """`srx_group` of include/srx.h: the exchange steps through the library itself (RCCL loaded by libsrx.so, row
    packing in its kernels), one rank per process with its GPU current. `unique_id()` on rank 0, hand the 128 bytes to
    the other ranks, then `Group(id, rank, nranks)`; `from_torch_dist()` does the hand-over with torch.distributed."""

    def __init__(self, id128: bytes, rank: int, nranks: int):
        import ctypes as C
        from . import _lib
        self.L, self.C, self._check = _lib.lib(), C, _lib.check
        self.h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id128), 128)
        self._check(self.L.srx_group_create(buf, C.c_int32(rank), C.c_int32(nranks), C.byref(self.h)), "srx_group_create")
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().srx_group_unique_id(buf), "srx_group_unique_id")
        return buf.raw

    @classmethod
    def from_torch_dist(cls):
        """Every rank of an initialised torch.distributed job calls this: rank 0's id travels by broadcast."""
        rank, world = (dist.get_rank(), dist.get_world_size()) if _is_dist() else (0, 1)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(box[0], rank, world)

    def nranks_seen(self) -> int:
        """The number of ranks the library's communicator was made for (`srx_group_rank`), as the library reports it."""
        C = self.C
        r, n = C.c_int32(-1), C.c_int32(-1)
        self._check(self.L.srx_group_rank(self.h, C.byref(r), C.byref(n)), "srx_group_rank")
        if r.value != self.rank:
            raise RuntimeError(f"srx_group_rank says rank {r.value}, this process is rank {self.rank}")
        return int(n.value)

    def close(self):
        if self.h:
            self.L.srx_group_destroy(self.h)
            self.h = self.C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ptr(self, t):
        if t is None:
            return None
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("the exchange takes contiguous device tensors")
        torch.cuda.current_stream(t.device).synchronize()
        return self.C.c_void_p(t.data_ptr())

    def gather_framebuffer(self, accum4, aux4, height, width, tile=TILE):
        C = self.C
        self._check(self.L.srx_group_gather_framebuffer(self.h, self._ptr(accum4), self._ptr(aux4), C.c_int32(width), C.c_int32(height),
                                                        C.c_int32(tile)), "srx_group_gather_framebuffer")

    def allgather_rows(self, frame, floats_per_pixel, height, width, tile=TILE):
        C = self.C
        self._check(self.L.srx_group_allgather_rows(self.h, self._ptr(frame), C.c_int32(floats_per_pixel), C.c_int32(width), C.c_int32(height),
                                                    C.c_int32(tile)), "srx_group_allgather_rows")

    def gather_framebuffer_dealt(self, deal, accum4, aux4):
        self._check(self.L.srx_group_gather_framebuffer_dealt(self.h, deal.h, self._ptr(accum4), self._ptr(aux4)), "srx_group_gather_framebuffer_dealt")

    def allgather_rows_dealt(self, deal, frame, floats_per_pixel):
        self._check(self.L.srx_group_allgather_rows_dealt(self.h, deal.h, self._ptr(frame), self.C.c_int32(floats_per_pixel)), "srx_group_allgather_rows_dealt")

    def allreduce(self, sums=None, mins=None):
        """In place: `sums` (float64 device tensor) summed over the ranks, `mins` (int32 device tensor) minimised."""
        C = self.C
        if sums is not None and sums.dtype != torch.float64 or mins is not None and mins.dtype != torch.int32:
            raise ValueError("sums must be float64, mins int32")
        self._check(self.L.srx_group_allreduce(self.h, self._ptr(sums), C.c_uint32(0 if sums is None else sums.numel()), self._ptr(mins),
                                               C.c_uint32(0 if mins is None else mins.numel())), "srx_group_allreduce")

    def complete_error_map(self, err, min_samples, height, width, tile=TILE):
        """The controller's exchange: every rank filled its own rows of the error map (`srx_noise_error_map`); all-gather
        them (each rank's rows travel once, no arithmetic: the complete map is bit-identical to one GPU's, so
        `analyze_noise` gives the same regions everywhere) and take the minimum sample count. Returns that minimum."""
        self.allgather_rows(err, 1, height, width, tile)
        mn = torch.tensor([int(min_samples)], dtype=torch.int32, device=err.device)
        self.allreduce(None, mn)
        return int(mn.item())


def rows_pack_unpack_roundtrip(frames_by_part, floats_per_pixel, height, width, tile=TILE):
    """What N ranks' gather does, played on ONE GPU through srx_rows_pack / srx_rows_unpack (tests): part p's frame
    holds valid rows of part p only; returns the frame every rank would end up with."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    n = len(frames_by_part)
    slot = int(L.srx_rows_slot_floats(C.c_int32(floats_per_pixel), C.c_int32(width), C.c_int32(height), C.c_int32(tile), C.c_int32(n)))
    slots = torch.full((n * slot,), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for p, f in enumerate(frames_by_part):
        _lib.check(L.srx_rows_pack(C.c_void_p(f.data_ptr()), C.c_int32(floats_per_pixel), C.c_int32(width), C.c_int32(height), C.c_int32(tile),
                                   C.c_int32(p), C.c_int32(n), C.c_void_p(slots.data_ptr() + 4 * p * slot), None), "srx_rows_pack")
    out = torch.full_like(frames_by_part[0], float("nan"))
    _lib.check(L.srx_rows_unpack(C.c_void_p(slots.data_ptr()), C.c_uint64(slot), C.c_uint64(0), C.c_int32(floats_per_pixel), C.c_int32(width),
                                 C.c_int32(height), C.c_int32(tile), C.c_int32(n), C.c_int32(-1), C.c_void_p(out.data_ptr()), None), "srx_rows_unpack")
    torch.cuda.synchronize()
    return out


def rows_pack_unpack_roundtrip_dealt(frames_by_part, floats_per_pixel, deal: Deal):
    """`rows_pack_unpack_roundtrip` under a deal (srx_rows_pack_dealt / srx_rows_unpack_dealt on ONE GPU)."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    n = len(frames_by_part)
    slot = int(L.srx_deal_slot_floats(deal.h, C.c_int32(floats_per_pixel)))
    slots = torch.full((n * slot,), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for p, f in enumerate(frames_by_part):
        _lib.check(L.srx_rows_pack_dealt(deal.h, C.c_void_p(f.data_ptr()), C.c_int32(floats_per_pixel), C.c_int32(p),
                                         C.c_void_p(slots.data_ptr() + 4 * p * slot), None), "srx_rows_pack_dealt")
    out = torch.full_like(frames_by_part[0], float("nan"))
    _lib.check(L.srx_rows_unpack_dealt(deal.h, C.c_void_p(slots.data_ptr()), C.c_uint64(slot), C.c_uint64(0), C.c_int32(floats_per_pixel), C.c_int32(-1),
                                       C.c_void_p(out.data_ptr()), None), "srx_rows_unpack_dealt")
    torch.cuda.synchronize()
    return out


def allreduce_noise_stats(sums: torch.Tensor, min_samples: torch.Tensor):
    """Combine adaptive-sampling statistics: `sums` (any shape, summed) and `min_samples` (minimum)."""
    if _is_dist():
        for t, op in ((sums, dist.ReduceOp.SUM), (min_samples, dist.ReduceOp.MIN)):
            _host_staged(lambda x, op=op: dist.all_reduce(x, op=op), t)
    return sums, min_samples


def allreduce_error_map(err: torch.Tensor, min_samples: int, height: int = 0, width: int = 0, tile: int = TILE):
    """Completes the per-pixel error map of the adaptive-sampling controller over torch.distributed (the transport of
    the CPU tests and rehearsals; on GPUs `Group.complete_error_map` does it through the library): every rank filled
    its own rows and left the others 0 (`srx_noise_error_map`). With the frame's shape given, the rows are ALL-GATHERED
    (each rank's rows travel once: 1/N of the map per rank, no arithmetic, so the map is bit-identical to one GPU's);
    without it they are summed with an all-reduce (twice the traffic; adding zeros is exact too). The minimum sample
    count is all-reduce(min) (`gather_thread_stats`, SceneManager.h:492-501). Returns the global minimum."""
    if not _is_dist():
        return int(min_samples)
    if height and width:
        gather_framebuffer_interleaved(err.view(height * width, 1), height, width, tile)
    else:
        _host_staged(lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM), err)
    mn = torch.tensor([int(min_samples)], dtype=torch.int64)
    if dist.get_backend() != "gloo":
        mn = mn.to(err.device)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    return int(mn.item())


def luminance(rgb: torch.Tensor):
    """`ImageProcessing::luminance` weights (ImageProcessing.h:115-118)."""
    return 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]


def rows_noise_stats(accum4: torch.Tensor, aux4: torch.Tensor, height: int, width: int, rows: torch.Tensor):
    """`slab_noise_stats` over an arbitrary set of owned rows."""
    rows = rows.to(accum4.device)
    return _noise_stats(accum4.view(height, width, 4)[rows], aux4.view(height, width, 4)[rows])


def slab_noise_stats(accum4: torch.Tensor, aux4: torch.Tensor, height: int, width: int, y0: int, y1: int):
    """Per-pixel error |L - L_aux| / max(L^1.1, L) (SceneManager.h:566-579) summed over this rank's slab,
    plus the pixel count and the minimum sample count. Returns (sums[2], min_samples[1])."""
    return _noise_stats(accum4.view(height, width, 4)[y0:y1], aux4.view(height, width, 4)[y0:y1])


def _noise_stats(a: torch.Tensor, x: torch.Tensor):
    L, La = luminance(a[..., :3]), luminance(x[..., :3])
    denom = torch.maximum(L.clamp_min(0) ** 1.1, L).clamp_min(1e-12)
    err = (L - La).abs() / denom
    sums = torch.stack([err.sum(), torch.tensor(float(err.numel()), device=err.device, dtype=err.dtype)])
    mins = a[..., 3].min().reshape(1) if a.numel() else torch.full((1,), float("inf"), device=a.device)
    return sums, mins
