"""The DEVICE's shading math against the reference's own object code, through srx_math_probe.

csrc/srx_probe.hip runs the inline helpers k_shade uses (csrc/srx_device.h) over arrays. They are compared bit for bit with
* oracle/_ref/libref_xyz.so      — XYZ.cpp as it lies: reflect, cross, normalize, dot, cdot, Quat::applyRotation, linear_mix,
                                   the `float - XYZ` quirk Q4;
* oracle/_ref/libref_matrix.so   — Matrix.h as it lies: Matrix3x3::applyTBNMatrix;
* oracle/_ref/ref_veclib_probe   — VecLib.cpp as it lies: full_sin, full_cos, tan;
built in the development container from /root/reference and carried to the GPU box as built files (oracle/_ref/ travels;
the reference's sources do not), and with the CPU oracle's restatement. A GPU whose divide, sqrt or FMA contraction differed
from x86's in any of these would show here, not only as a shading tolerance."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def _inputs(n=200_000):
    rng = np.random.default_rng(101)
    a, b, c = (rng.normal(size=(n, 3)).astype(np.float32) for _ in range(3))
    q = rng.normal(size=(n, 4)).astype(np.float32)
    a[:64] *= np.float32(1e-18); b[64:128] *= np.float32(1e15); c[128:192] *= np.float32(1e-25)
    a[192:256, 0] = 0.0; b[256:320] = 0.0
    a[1000:, 0] = rng.uniform(-2 * np.pi, 4 * np.pi, n - 1000).astype(np.float32)      # a.x doubles as the trig argument
    return a, b, c, q


def _same(x, y):
    return (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))


def _probe(a, b, c, q, s):
    from raytracing_amd import _lib
    L = _lib.lib()
    _lib.check(L.srx_init(C.c_int(0)), "srx_init")
    out = np.zeros((len(a), 26), np.float32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    _lib.check(L.srx_math_probe(p(a), p(b), p(c), p(q), C.c_float(s), C.c_uint32(len(a)), p(out)), "srx_math_probe")
    return out


def test_device_math_equals_the_oracle():
    from oracle import sro
    a, b, c, q = _inputs()
    n = len(a)
    s = np.float32(0.37)
    dev = _probe(a, b, c, q, s)
    L = sro.lib()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    r3, c3, n3 = (np.zeros((n, 3), np.float32) for _ in range(3)); d1 = np.zeros(n, np.float32)
    L.sro_vec_ops(p(a), p(b), C.c_uint32(n), p(r3), p(c3), p(n3), p(d1))
    rot = np.zeros((n, 3), np.float32)
    L.sro_apply_rotation(p(a), p(q), C.c_uint32(n), p(rot))
    f3, m3, s3 = (np.zeros((n, 3), np.float32) for _ in range(3)); cd = np.zeros(n, np.float32)
    L.sro_misc(p(a), p(b), C.c_float(s), C.c_uint32(n), p(f3), p(m3), p(cd), p(s3))
    tbn = np.zeros((n, 3), np.float32)
    qn = np.ascontiguousarray(q[:, :3])
    L.sro_apply_tbn(p(a), p(b), p(c), p(qn), C.c_uint32(n), p(tbn))
    si, co, ta = sro.trig(np.ascontiguousarray(a[:, 0]))
    for name, got, want in (("reflect", dev[:, 0:3], r3), ("cross", dev[:, 3:6], c3), ("normalize", dev[:, 6:9], n3), ("dot", dev[:, 9], d1),
                            ("cdot", dev[:, 10], cd), ("applyRotation", dev[:, 11:14], rot), ("linear_mix", dev[:, 14:17], m3),
                            ("applyTBNMatrix", dev[:, 17:20], tbn), ("full_sin", dev[:, 20], si), ("full_cos", dev[:, 21], co),
                            ("tan", dev[:, 22], ta), ("float - XYZ (Q4)", dev[:, 23:26], f3)):
        ok = _same(np.ascontiguousarray(got), np.ascontiguousarray(want))
        assert ok.all(), (name, int((~ok).sum()))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "libref_xyz.so")) or not os.path.exists(os.path.join(REF_DIR, "libref_matrix.so")),
                    reason="oracle/_ref not built (made in the development container, where /root/reference exists)")
def test_device_math_equals_reference_object_code(tmp_path):
    a, b, c, q = _inputs()
    n = len(a)
    s = np.float32(0.37)
    dev = _probe(a, b, c, q, s)
    X = C.CDLL(os.path.join(REF_DIR, "libref_xyz.so"))
    M = C.CDLL(os.path.join(REF_DIR, "libref_matrix.so"))
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    r3, c3, n3 = (np.zeros((n, 3), np.float32) for _ in range(3)); d1 = np.zeros(n, np.float32)
    X.ref_vec_ops(p(a), p(b), C.c_uint32(n), p(r3), p(c3), p(n3), p(d1))
    rot = np.zeros((n, 3), np.float32)
    X.ref_apply_rotation(p(a), p(q), C.c_uint32(n), p(rot))
    f3, m3, s3 = (np.zeros((n, 3), np.float32) for _ in range(3)); cd = np.zeros(n, np.float32)
    X.ref_misc(p(a), p(b), C.c_float(s), C.c_uint32(n), p(f3), p(m3), p(cd), p(s3))
    tbn = np.zeros((n, 3), np.float32)
    qn = np.ascontiguousarray(q[:, :3])
    M.ref_apply_tbn(p(a), p(b), p(c), p(qn), C.c_uint32(n), p(tbn))
    checks = [("reflect", dev[:, 0:3], r3), ("cross", dev[:, 3:6], c3), ("normalize", dev[:, 6:9], n3), ("dot", dev[:, 9], d1),
              ("cdot", dev[:, 10], cd), ("applyRotation", dev[:, 11:14], rot), ("linear_mix", dev[:, 14:17], m3),
              ("applyTBNMatrix", dev[:, 17:20], tbn), ("float - XYZ (Q4)", dev[:, 23:26], f3)]
    probe = os.path.join(REF_DIR, "ref_veclib_probe")
    if os.path.exists(probe):
        x = np.ascontiguousarray(a[:, 0])
        src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
        with open(src, "wb") as fh:
            fh.write(np.array([n, 0], np.uint32).tobytes()); fh.write(x.tobytes())
        subprocess.run([probe, str(src), str(dst)], check=True, timeout=120)
        trig = np.fromfile(dst, np.float32)[: 3 * n].reshape(n, 3)
        checks += [("full_sin", dev[:, 20], trig[:, 0]), ("full_cos", dev[:, 21], trig[:, 1]), ("tan", dev[:, 22], trig[:, 2])]
    for name, got, want in checks:
        ok = _same(np.ascontiguousarray(got), np.ascontiguousarray(want))
        assert ok.all(), (name, int((~ok).sum()))
