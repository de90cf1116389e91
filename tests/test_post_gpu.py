"""Output stage on the GPU through the C-ABI against the oracle.
Tolerance: the stage is float/double arithmetic restated operation by operation; the one libm call on the parity path
is pow(double, double), where the device library and glibc may differ in the last place of the DOUBLE result, which
reaches the float result only on a rounding tie. Bar: float outputs equal within 1 ulp (2e-5 absolute on the 0..255
scale), 8-bit canvas identical except where the float sits within that distance of an integer."""
import struct
import zlib

import numpy as np
import pytest
import torch

from oracle import sro
from post_inputs import hdr_pixels
from raytracing_amd import _lib, output

pytestmark = pytest.mark.gpu


def _frame(res_x, res_y, seed=5):
    px = hdr_pixels(res_x * res_y, seed)
    a4 = np.concatenate([px, np.full((len(px), 1), 7, np.float32)], axis=1)
    return px, torch.from_numpy(a4).cuda().contiguous()


@pytest.mark.parametrize("tonemap", [output.TONEMAP_NONE, output.TONEMAP_AGX_APPROX])
def test_post_process_matches_oracle(tonemap):
    rx, ry = 203, 157                                   # not a multiple of the block size
    px, acc = _frame(rx, ry)
    rgba, rgb = output.post_process_raw(acc, rx, ry, tonemap, want_float=True)
    f = rgb.cpu().numpy().reshape(-1, 3)
    u8 = rgba.cpu().numpy().reshape(-1, 4)
    wf, wu8 = sro.post_process(px, tonemap)
    assert np.array_equal(np.isnan(f), np.isnan(wf))
    fin = ~np.isnan(wf)
    tol = 2e-5 if tonemap == output.TONEMAP_NONE else 5e-3   # AgX stand-in: log2f + a 6th-order polynomial in float
    assert np.abs(f[fin] - wf[fin]).max() <= tol
    if tonemap == output.TONEMAP_NONE:
        assert (f[fin] != wf[fin]).mean() < 1e-3          # all but rounding ties are bit-identical
    diff = u8[:, :3].astype(int) - wu8[:, :3].astype(int)
    assert np.abs(diff).max() <= 1 and (u8[:, 3] == 255).all()
    near_integer = np.abs(wf - np.round(wf)) <= tol
    assert not (diff != 0)[~near_integer & fin].any()
    assert np.array_equal(u8[:, :3], np.where(np.isnan(f), 0, f).astype(np.int32).astype(np.uint8))   # truncation of its own floats


def test_raw_and_png_files(tmp_path):
    rx, ry = 131, 67
    px, acc = _frame(rx, ry, seed=9)
    output.write_raw_file(acc, rx, ry, tmp_path / "p.raw")
    sro.write_raw(tmp_path / "o.raw", px, rx, ry)
    assert open(tmp_path / "p.raw", "rb").read() == open(tmp_path / "o.raw", "rb").read()    # byte-identical dump
    back, bx, by = output.read_raw(tmp_path / "p.raw")
    assert (bx, by) == (rx, ry) and np.array_equal(back.reshape(-1, 3).view(np.uint32), px.view(np.uint32))
    rgba = output.post_process_raw(acc, rx, ry)
    output.write_png(rgba, tmp_path / "c.png")
    b = open(tmp_path / "c.png", "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, ihdr = 8, b"", None
    while at < len(b):
        n, typ = struct.unpack(">I4s", b[at:at + 8])
        body = b[at + 8:at + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", b[at + 8 + n:at + 12 + n])[0]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        if typ == b"IDAT":
            idat += body
        at += 12 + n
    assert ihdr == (rx, ry, 8, 6, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(ry, 1 + 4 * rx)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(ry, rx, 4), rgba.cpu().numpy())


def test_full_4k_frame_properties():
    """BASELINE's largest frame (3840x2160): size-independent properties instead of a CPU comparison of every pixel."""
    rx, ry = 3840, 2160
    g = torch.Generator(device="cuda").manual_seed(1)
    acc = torch.rand((ry * rx, 4), device="cuda", generator=g) * 2
    rgba, rgb = output.post_process_raw(acc, rx, ry, want_float=True)
    assert rgba.shape == (ry, rx, 4) and bool((rgba[..., 3] == 255).all())
    assert bool(torch.equal(rgb.to(torch.int32).to(torch.uint8), rgba[..., :3]))        # canvas = truncation of the floats
    # scaling by one factor per pixel keeps the largest channel the largest
    a = acc[:, :3].reshape(ry, rx, 3)
    srt = a.sort(dim=2).values
    clear = (rgb.amax(dim=2) < 255) & (srt[..., 2] > 1.01 * srt[..., 1])
    assert bool(torch.equal(a.argmax(dim=2)[clear], rgb.argmax(dim=2)[clear]))
    # a strip checked against the oracle
    idx = torch.arange(0, ry * rx, 4099, device="cuda")
    wf, _ = sro.post_process(acc[idx, :3].cpu().numpy(), 0)
    assert np.abs(rgb.reshape(-1, 3)[idx].cpu().numpy() - wf).max() <= 2e-5
    gray = torch.linspace(0, 1, 4096, device="cuda").repeat_interleave(3).reshape(-1, 3)
    acc2 = torch.cat([gray, torch.ones((4096, 1), device="cuda")], dim=1).contiguous()
    _, r2 = output.post_process_raw(acc2, 64, 64, want_float=True)
    v = r2.reshape(-1, 3)[:, 0]
    assert bool((v[1:] >= v[:-1]).all())                                                 # a gray ramp stays monotone


def test_bad_arguments_fail_loudly():
    acc = torch.zeros((16, 4), device="cuda")
    with pytest.raises(_lib.SrxError):
        output.post_process_raw(acc, 4, 4, tonemap=9)
    with pytest.raises(_lib.SrxError):
        output.post_process_raw(acc, 5, 4)
    with pytest.raises(_lib.SrxError):
        output.write_raw_file(acc, 4, 4, "/nonexistent_dir/x.raw")
