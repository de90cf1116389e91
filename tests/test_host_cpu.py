"""CPU checks of the product's host side (no GPU, no compute kernels): packager parity with the oracle,
table builder parity, C-ABI symbol coverage and error behaviour, tile sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import sro
from raytracing_amd import _lib, engine, scenes, tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "srx.h")).read()
    declared = sorted(set(re.findall(r"\b(srx_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 19
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_lib.EXPORTS) == declared


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(16, 1),
    lambda: scenes.soup(3000, seed=4),
    lambda: scenes.atrium(12000, res=(32, 18), subdiv=1, tex_size=16),
    lambda: scenes.interior(5000, res=(32, 18), subdiv=1, tex_size=16),
    lambda: scenes.diffuse_box(2000, res=(32, 18), subdiv=1),
])
def test_product_packager_matches_oracle_bytes(make):
    desc = make()
    PS = engine.package(desc)
    S = sro.OracleScene(desc, build_tables=False)
    n, pk, tr = S.packaged()
    assert np.array_equal(n, PS.avx_bvh) and np.array_equal(pk, PS.avx_tri_data) and np.array_equal(tr, PS.tri_data)


def test_counts_of_the_2000_triangle_soup():
    """DESIGN.md section 2 cites them: the 2 000-triangle soup packages to 73 nodes and 250 packets in both
    implementations — the counts SURVEY 8(c) recorded for a 2 000-triangle scene built by the reference itself (that
    scene's triangles were not kept, so this pins a count, not bytes: 250 = every leaf full, 73 = the 8-wide collapse
    of a median-split tree over 250 leaves)."""
    desc = scenes.soup(2000, seed=3)
    PS = engine.package(desc)
    S = sro.OracleScene(desc, build_tables=False)
    assert (len(PS.avx_bvh), len(PS.avx_tri_data), len(PS.tri_data)) == (73, 250, 2000)
    assert S.counts() == (73, 250, 2000)


def test_degenerate_inputs():
    """Ragged/edge inputs of the packager: one triangle, 9 coincident triangles (degenerate split, a leaf with
    more than 8 triangles -> 2 packets), and zero triangles (the reference asserts, BVH.h:100)."""
    one = scenes.soup(1, seed=1)
    PS = engine.package(one)
    assert PS.avx_bvh.shape[0] == 1 and PS.avx_tri_data.shape[0] == 1 and PS.tri_data.shape[0] == 1
    nine = scenes.soup(1, seed=1)
    nine.tris = np.repeat(nine.tris, 9)
    PS = engine.package(nine)
    S = sro.OracleScene(nine, build_tables=False)
    n, pk, tr = S.packaged()
    assert np.array_equal(n, PS.avx_bvh) and np.array_equal(pk, PS.avx_tri_data)
    assert PS.avx_tri_data.shape[0] == 2 and PS.avx_bvh[0, 192:196].view(np.uint32)[0] == 2
    empty = scenes.soup(1, seed=1)
    empty.tris = empty.tris[:0]
    with pytest.raises(_lib.SrxError):
        engine.package(empty)


def test_tables_match_oracle(tables):
    a, p = engine.build_tables()
    assert np.array_equal(a.view(np.uint32), tables[0].view(np.uint32))
    assert np.array_equal(p.view(np.uint32), tables[1].view(np.uint32))


def test_no_gpu_fails_loudly():
    """Without a HIP device the library must refuse, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SrxError):
        engine.RayEngine(0)


def test_slabs_partition_the_frame():
    for H in (1080, 2160, 37, 4):
        for world in (1, 2, 3, 4, 8):
            cuts = tiles.slabs(world, H)
            assert cuts[0][0] == 0 and cuts[-1][1] == H
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(y0 % tiles.TILE == 0 for y0, _ in cuts)


def test_cpp_adapter_header_compiles():
    """include/spatialray_adapter.hpp (the reference-side binding of INTEGRATION.md) must compile as plain C++17."""
    import subprocess
    src = '#include "spatialray_adapter.hpp"\nint main() { spatialray::RayEngine* r = nullptr; (void)r; return 0; }\n'
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-"],
                   input=src.encode(), check=True)


def test_parallel_packager_is_thread_count_independent(monkeypatch):
    """Subtrees are built by several threads above 4 096 triangles; the package must not depend on the crew size and
    must still equal the oracle's (serial) packager byte for byte."""
    from oracle import sro
    from raytracing_amd import engine, scenes
    desc = scenes.soup(12000, seed=21)
    out = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("SRX_BUILD_THREADS", threads)
        PS = engine.package(desc)
        out.append((PS.avx_bvh.tobytes(), PS.avx_tri_data.tobytes(), PS.tri_data.tobytes()))
    assert out[0] == out[1] == out[2]
    S = sro.OracleScene(desc, build_tables=False)
    n, p, t = S.packaged()
    assert out[0] == (n.tobytes(), p.tobytes(), t.tobytes())


def test_trace_dealing_tiles_the_ray_range():
    """The traversal kernel deals its rays to blocks in full portions followed by a shrinking tail (srx_trace_layout in
    csrc/srx_kernels.h, the same function on host and device). Whatever the sizes: the block ranges tile [0, n) in order,
    no block is empty, and the grid bound the launcher takes covers every smaller ray count (launches that read their count
    on the device)."""
    import ctypes as C
    L = _lib.lib()
    rng = np.random.default_rng(3)
    sizes = [0, 1, 7, 8, 9, 63, 64, 65, 511, 512, 513, 4095, 65536, 100001, 2073600] + [int(x) for x in rng.integers(1, 3_000_000, 12)]
    for k in (1, 2, 4, 8):
        for tail in (0, 1, 5, 8, 24, 8192):
            for n in sizes:
                nb, bound = C.c_uint32(), C.c_uint32()
                assert L.srx_trace_dealing(C.c_uint32(n), C.c_int32(k), C.c_uint32(tail), C.c_uint32(0), None, None, C.byref(nb), C.byref(bound)) == 0
                start, count = np.zeros(nb.value, np.uint32), np.zeros(nb.value, np.uint32)
                p = lambda a: a.ctypes.data_as(C.c_void_p)
                L.srx_trace_dealing(C.c_uint32(n), C.c_int32(k), C.c_uint32(tail), C.c_uint32(nb.value), p(start), p(count), None, None)
                assert (count > 0).all() and (count <= 8 * k).all()
                assert np.array_equal(start.astype(np.int64), np.concatenate([[0], np.cumsum(count.astype(np.int64))[:-1]]).astype(np.int64)[:len(start)]) if nb.value else n == 0
                assert int(count.sum()) == n
                assert nb.value <= bound.value
                if tail and nb.value > 1:
                    assert (np.diff(count.astype(np.int64)[:-1]) <= 0).all()        # sizes never grow along the range (last block may be partial)
    for k in (2, 4, 8):                                   # the bound holds for every smaller count
        for tail in (0, 3, 8, 16):
            for n in list(range(0, 700)) + [5000, 5001, 5063, 5064]:
                bound = C.c_uint32()
                L.srx_trace_dealing(C.c_uint32(n), C.c_int32(k), C.c_uint32(tail), C.c_uint32(0), None, None, None, C.byref(bound))
                for m in range(max(0, n - 130), n + 1):
                    nb = C.c_uint32()
                    L.srx_trace_dealing(C.c_uint32(m), C.c_int32(k), C.c_uint32(tail), C.c_uint32(0), None, None, C.byref(nb), None)
                    assert nb.value <= bound.value, (k, tail, n, m)
    assert L.srx_trace_dealing(C.c_uint32(10), C.c_int32(3), C.c_uint32(0), C.c_uint32(0), None, None, None, None) != 0


def test_xcd_map_is_a_permutation_that_keeps_chunks_on_one_xcd():
    import ctypes as C
    L = _lib.lib()
    for nblk in (8, 16, 64, 72, 1000 * 8, 4104, 25232 + 8):
        for chunk in (1, 3, 8, 32, 512, 0x7fffffff):
            out = np.zeros(nblk, np.uint32)
            assert L.srx_trace_xcd_map(C.c_uint32(nblk), C.c_uint32(chunk), out.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(np.sort(out), np.arange(nblk, dtype=np.uint32)), (nblk, chunk)
            per = nblk // 8
            full = (per // chunk) * chunk * 8                       # logical blocks that lie in whole chunks
            lb = out.astype(np.int64)
            hw_xcd = np.arange(nblk) % 8
            inside = lb < full
            assert np.array_equal((lb[inside] // chunk) % 8, hw_xcd[inside])     # chunk c runs on XCD c % 8
            # each XCD sees its logical blocks in ascending order (blocks are dispatched in index order)
            for x in range(8):
                assert (np.diff(lb[hw_xcd == x]) > 0).all()
    assert L.srx_trace_xcd_map(C.c_uint32(12), C.c_uint32(4), np.zeros(12, np.uint32).ctypes.data_as(C.c_void_p)) != 0
