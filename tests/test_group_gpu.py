"""The multi-GPU exchange of include/srx.h (srx_group_*, srx_rows_*) as far as one GPU can show it.

* The row packing kernels play an N-rank gather on one GPU: part p's frame holds only part p's rows; after pack (every
  part) + unpack the frame is the composition of all parts, for ragged heights and for 1 and 4 floats per pixel.
* A communicator of ONE rank goes through the real RCCL calls (ncclCommInitRank, ncclAllGather, ncclAllReduce loaded by
  libsrx.so): gather and all-reduce must leave a single rank's data as it is.
* examples/render_glb with --rank 0 --nranks 1 --id-file renders through spatialray::Group and must write the frame the
  plain run writes.
More than one rank needs more than one GPU: UNMEASURED ON HARDWARE until the driver's scaling run; the N > 1 control flow
is covered by the gloo world-2 tests of tests/test_multigpu_cpu.py.
"""
import os
import subprocess

import numpy as np
import pytest

from gltf_files import write_test_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("H,W,fpp,n", [(22, 7, 1, 3), (1080, 64, 4, 8), (5, 3, 4, 2), (36, 16, 1, 1)])
def test_rows_pack_unpack_compose_the_frame(H, W, fpp, n):
    import torch
    from raytracing_amd import tiles
    full = torch.rand(H, W, fpp, device="cuda")
    parts = []
    for p in range(n):
        f = torch.full_like(full, -7.0)                   # garbage where the part has no rows
        rows = tiles.interleaved_rows(p, n, H).cuda()
        f[rows] = full[rows]
        parts.append(f.contiguous())
    out = tiles.rows_pack_unpack_roundtrip(parts, fpp, H, W)
    assert torch.equal(out, full)


def test_one_rank_group_through_rccl():
    import torch
    from raytracing_amd import engine, tiles
    engine.RayEngine(0)
    g = tiles.Group(tiles.Group.unique_id(), 0, 1)
    H, W = 37, 19
    acc = torch.rand(H * W, 4, device="cuda"); aux = torch.rand(H * W, 4, device="cuda")
    a0, x0 = acc.clone(), aux.clone()
    g.gather_framebuffer(acc, aux, H, W)
    g.gather_framebuffer(acc, None, H, W)
    assert torch.equal(acc, a0) and torch.equal(aux, x0)
    err = torch.rand(H * W, device="cuda"); e0 = err.clone()
    assert g.complete_error_map(err, 41, H, W) == 41 and torch.equal(err, e0)
    s = torch.tensor([1.5, 2.5, -3.0], dtype=torch.float64, device="cuda"); m = torch.tensor([7, -2], dtype=torch.int32, device="cuda")
    g.allreduce(s, m)
    assert s.tolist() == [1.5, 2.5, -3.0] and m.tolist() == [7, -2]
    g.close()


def test_cpp_host_renders_through_a_group(tmp_path):
    exe = os.path.join(ROOT, "examples", "render_glb")
    assert os.path.exists(exe), "examples/render_glb is missing: run __graft_entry__.build()"
    glb = tmp_path / "scene.glb"
    write_test_scene(str(glb), seed=3, textures=True)
    args = [str(glb), "64", "36", "2", "3", "77"]
    (tmp_path / "id").write_bytes(b"\x55" * 128)          # an id left by a job that died: rank 0 removes it before anything else
    plain = subprocess.run([exe] + args + [str(tmp_path / "a.raw")], capture_output=True, text=True, timeout=120)
    grouped = subprocess.run([exe] + args + [str(tmp_path / "b.raw"), "--rank", "0", "--nranks", "1", "--id-file", str(tmp_path / "id")],
                             capture_output=True, text=True, timeout=120)
    assert plain.returncode == 0, plain.stderr
    assert grouped.returncode == 0, grouped.stderr
    last = lambda out: out.strip().splitlines()[-1].split(":")[1].split(",")[:2]       # (RCCL prints a version banner first)
    assert last(plain.stdout) == last(grouped.stdout)                                   # the same rays and shadow rays
    a = np.fromfile(tmp_path / "a.raw", np.float32); b = np.fromfile(tmp_path / "b.raw", np.float32)
    assert a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=1e-6)
    assert not (tmp_path / "id").exists()                  # removed again once every rank has joined
