"""Output stage on the CPU: the oracle (oracle/sro_post.h) against known answers and its committed
vectors, and the product's .raw reader (host-only C-ABI call) against the oracle's writer. PARITY UNPINNED upstream:
the reference's ImageHandler cannot be built here (OpenColorIO + a config outside its repository) and holds no fixture."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import sro
from post_inputs import hdr_pixels
from raytracing_amd import _lib, output

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_known_answers():
    px = np.array([[0, 0, 0], [0.5, 0.5, 0.5], [1, 0, 0], [4, 3, 2], [-1, -1, -1], [np.nan, 0.5, 0.5]], np.float32)
    f, u8 = sro.post_process(px, 0)
    assert np.array_equal(u8[0], [0, 0, 0, 255])                      # lums == 0 -> (0, 0, 0) (ImageProcessing.h:130)
    lum = np.float32(0.5) * np.float32(0.2126) + np.float32(0.5) * np.float32(0.7152) + np.float32(0.5) * np.float32(0.0722)
    g = np.float32(np.float32(0.5) * np.float32(float(lum) ** (1.0 / float(np.float32(2.2)))) / lum)
    assert abs(f[1, 0] - min(1.0, float(g)) * 255) < 1e-3 and f[1, 0] == f[1, 1] == f[1, 2]   # gray stays gray (:25-30)
    assert np.array_equal(u8[1, :3], [186, 186, 186])                 # truncation, not rounding (186.08)
    assert np.array_equal(u8[2, :3], [255, 0, 0]) and np.array_equal(u8[3, :3], [255, 255, 255])
    assert np.isnan(f[4]).all() and np.array_equal(u8[4, :3], [0, 0, 0])   # pow(negative, 1/2.2) = NaN -> stored as 0
    assert np.isnan(f[5]).all()


def test_hue_is_kept_and_values_are_in_range():
    px = hdr_pixels(4096)
    f, u8 = sro.post_process(px, 0)
    ok = np.isfinite(f).all(axis=1)
    assert ok.sum() > 4000
    assert (f[ok] >= 0).all() and (f[ok] <= 255).all()
    assert np.array_equal(u8[ok, :3], f[ok].astype(np.int32).astype(np.uint8))
    # the stage scales a pixel by one factor (before clamping): a clearly largest channel stays the largest
    srt = np.sort(px, axis=1)
    pos = ok & (px > 0).all(axis=1) & (srt[:, 2] > 1.01 * srt[:, 1]) & (f.max(axis=1) < 255)
    assert pos.sum() > 1000 and np.array_equal(np.argmax(px[pos], axis=1), np.argmax(f[pos], axis=1))


def test_oracle_matches_committed_vectors():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "post_golden.json")))
    px = np.array(g["pixels_u32"], np.uint32).view(np.float32)
    assert np.array_equal(px.view(np.uint32), hdr_pixels(64).view(np.uint32))
    for tm in (0, 1):
        f, u8 = sro.post_process(px, tm)
        want = np.array(g[f"float_u32_tonemap{tm}"], np.uint32)
        if tm == 0:
            assert np.array_equal(f.view(np.uint32), want)
        else:   # log2f of the host libm may differ in the last place between machines
            w = want.view(np.float32)
            fin = np.isfinite(w)
            assert np.array_equal(np.isfinite(f), fin) and np.allclose(f[fin], w[fin], rtol=0, atol=2e-3)
        assert np.abs(u8.astype(int) - np.array(g[f"rgba8_tonemap{tm}"], int)).max() <= (0 if tm == 0 else 1)


def test_raw_file_format_and_product_reader(tmp_path):
    rng = np.random.default_rng(3)
    rx, ry = 37, 11
    img = rng.uniform(0, 4, (ry, rx, 3)).astype(np.float32)
    p = tmp_path / "a.raw"
    sro.write_raw(p, img, rx, ry)
    b = open(p, "rb").read()
    assert len(b) == 8 + rx * ry * 12 and struct.unpack("<II", b[:8]) == (rx, ry)          # FileLoading.h:172-181
    assert np.array_equal(np.frombuffer(b[8:8 + 12], np.float32), img[0, 0])                 # row 0, pixel 0 first
    back, bx, by = sro.read_raw(p)
    assert (bx, by) == (rx, ry) and np.array_equal(back, img)
    got, gx, gy = output.read_raw(p)                                                         # product, host-only call
    assert (gx, gy) == (rx, ry) and np.array_equal(got, img)
    with pytest.raises(_lib.SrxError):
        output.read_raw(tmp_path / "missing.raw")
    open(tmp_path / "short.raw", "wb").write(b[:100])
    with pytest.raises(_lib.SrxError):
        output.read_raw(tmp_path / "short.raw")
    open(tmp_path / "bad.raw", "wb").write(struct.pack("<ii", -3, 5))
    with pytest.raises(_lib.SrxError):
        output.read_raw(tmp_path / "bad.raw")
