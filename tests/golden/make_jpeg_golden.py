#!/usr/bin/env python3
"""Writes tests/golden/jpeg_cases.npz: small JPEG (and two PNG) files made with Pillow, each with the RGBA bytes the
reference's own image decoder gives for it (stb_image.h as vendored by the reference, compiled by oracle/ref_build into
oracle/_ref/libref_stb.so and asked for 4 components, as tinygltf does). Run in the build container, where
/root/reference exists; the .npz is data (image files + expected pixels) and travels with the repo.

Cases: 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2), 4:4:0 (h1v2), 4:1:1 (h4v1: stb's nearest-neighbour path), grayscale, odd sizes
down to 1x1, restart intervals, a 16-bit quantisation table (quality 1), an Adobe-marked RGB file (no colour transform),
progressive files (spectral selection + successive approximation, as libjpeg writes them; all three chroma layouts,
grayscale, restart intervals), and PNGs of every colour type and bit depth, plain and Adam7-interlaced, with tRNS."""
import ctypes as C
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_stb.so"))


def stb(data):
    buf = np.frombuffer(data, np.uint8)
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    if R.ref_stb_decode(buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(w), C.byref(h), C.byref(n), None):
        return None
    out = np.zeros((h.value, w.value, 4), np.uint8)
    R.ref_stb_decode(buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(w), C.byref(h), C.byref(n), out.ctypes.data_as(C.c_void_p))
    return out


def picture(w, h, seed, gray=False):
    """Smooth gradients + a sharp-edged checker + noise: exercises DC prediction, long AC runs and clamping."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 127 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0), 255.0 * x / max(w - 1, 1), 255.0 * ((x // 4 + y // 3) % 2)], -1)
    img = img + rng.normal(0, 20, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 0] if gray else img


def jpeg(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def resample(data, old, new):
    data = bytearray(data)
    i = data.index(b"\xff\xc0")
    assert data[i + 9] == 3 and data[i + 11] == old          # three components, the first with the expected factors
    data[i + 11] = new
    return bytes(data)


def main():
    cases = {}
    sub = {"444": 0, "422": 1, "420": 2}
    k = 0
    for name, s in sub.items():
        for (w, h) in ((33, 17), (16, 16), (1, 1), (7, 50), (64, 9)):
            k += 1
            cases[f"{name}_{w}x{h}"] = jpeg(picture(w, h, k), quality=60 + 5 * (k % 7), subsampling=s)
    cases["gray_21x13"] = jpeg(picture(21, 13, 40, gray=True), quality=80)
    cases["gray_8x8"] = jpeg(picture(8, 8, 41, gray=True), quality=95)
    cases["q1_420_40x24"] = jpeg(picture(40, 24, 42), quality=1, subsampling=2)          # 16-bit-free but coarse tables
    cases["q100_444_19x19"] = jpeg(picture(19, 19, 43), quality=100, subsampling=0)
    cases["restart_420_50x35"] = jpeg(picture(50, 35, 44), quality=70, subsampling=2, restart_marker_blocks=3)
    cases["restart_444_23x9"] = jpeg(picture(23, 9, 45), quality=70, subsampling=0, restart_marker_blocks=1)
    cases["opt_422_45x31"] = jpeg(picture(45, 31, 46), quality=85, subsampling=1, optimize=True)
    # Pillow writes no h1v2 / h4v1 files: the luma sampling byte of the frame header of a 32x32 file is rewritten instead
    # (same blocks per MCU and the same number of MCUs, so the entropy-coded data stays valid; the picture is scrambled,
    # which does not matter here)
    cases["h1v2_32x32"] = resample(jpeg(picture(32, 32, 47), quality=75, subsampling=1), 0x21, 0x12)
    cases["h4v1_32x32"] = resample(jpeg(picture(32, 32, 48), quality=75, subsampling=2), 0x22, 0x41)
    cases["rgb_adobe_20x14"] = jpeg(picture(20, 14, 49), quality=90, subsampling=0, keep_rgb=True)
    cases["big_420_200x120"] = jpeg(picture(200, 120, 50), quality=75, subsampling=2)
    cases["progressive_24x24"] = jpeg(picture(24, 24, 51), quality=75, progressive=True)
    cases["progressive_444_31x19"] = jpeg(picture(31, 19, 52), quality=85, progressive=True, subsampling=0)
    cases["progressive_422_40x21"] = jpeg(picture(40, 21, 53), quality=60, progressive=True, subsampling=1, optimize=True)
    cases["progressive_420_57x33"] = jpeg(picture(57, 33, 54), quality=35, progressive=True, subsampling=2, restart_marker_blocks=2)
    cases["progressive_gray_17x29"] = jpeg(picture(17, 29, 55, gray=True), quality=90, progressive=True)
    cases["progressive_q5_64x48"] = jpeg(picture(64, 48, 56), quality=5, progressive=True, subsampling=2)
    out = {}
    for name, data in cases.items():
        want = stb(data)
        assert want is not None, name
        out["file_" + name] = np.frombuffer(data, np.uint8)
        out["rgba_" + name] = want
        print(f"{name:22s} {len(data):6d} bytes -> {want.shape}")
    # PNG through the same decoder (tests/gltf_files.py writes them; the expected bytes come from stb_image too)
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from gltf_files import png_bytes
    rng = np.random.default_rng(9)
    for c in (1, 3, 4):
        img = rng.integers(0, 256, (10, 13, c), dtype=np.uint8)
        data = png_bytes(img)
        out[f"file_png{c}_13x10"] = np.frombuffer(data, np.uint8)
        out[f"rgba_png{c}_13x10"] = stb(data)
    # every colour type x bit depth, plain and Adam7, with and without tRNS (tests/gltf_files.py: png_bytes_general)
    from gltf_files import png_bytes_general
    rng = np.random.default_rng(10)
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for interlace in (False, True):
                c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
                w, h = int(rng.integers(3, 20)), int(rng.integers(3, 14))
                plte = trns = None
                if ctype == 3:
                    npal = min(1 << depth, int(rng.integers(2, 200)))
                    plte = rng.integers(0, 256, npal * 3).astype(np.uint8)
                    smp = rng.integers(0, npal, (h, w, 1))
                    if interlace:
                        trns = rng.integers(0, 256, max(1, npal // 2)).astype(np.uint8)
                else:
                    smp = rng.integers(0, 1 << depth, (h, w, c))
                    if ctype in (0, 2) and not interlace:
                        trns = b"".join(int(v).to_bytes(2, "big") for v in smp[h // 2, w // 2])
                data = png_bytes_general(smp, depth, ctype, interlace, plte, trns)
                name = f"pngx_c{ctype}_d{depth}_{'adam7' if interlace else 'plain'}_{w}x{h}"
                want = stb(data)
                assert want is not None, name
                out["file_" + name] = np.frombuffer(data, np.uint8)
                out["rgba_" + name] = want
    np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **out)
    print("wrote", os.path.join(HERE, "jpeg_cases.npz"), os.path.getsize(os.path.join(HERE, "jpeg_cases.npz")), "bytes")


if __name__ == "__main__":
    main()
