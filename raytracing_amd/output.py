"""Output stage: host-side mirror of the reference's `ImageHandler` / `FileManager` members for
the framebuffer, over the C-ABI of include/srx.h. Framebuffers and canvases are device tensors; there is no CPU path.

  post_process_raw  <- ImageHandler::postProcessRaw / post_process_pixel (ImageProcessing.h:137-158) + the 8-bit store of
                       GUI::commit_pixel (GUI.h:104-107), i.e. what SceneManager::refresh_canvas shows (SceneManager.h:1064-1072)
  write_raw_file    <- FileManager::writeRawFile (FileLoading.h:166-182)
  read_raw          <- FileManager::readRaw (FileLoading.h:183-199)
  write_png         -- the canvas as a file (the reference only shows it in a window)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

TONEMAP_NONE = 0          # the OpenColorIO step left out (parity path)
TONEMAP_AGX_APPROX = 1    # analytic AgX approximation in place of the reference's OCIO "AGX Base sRGB" view


def _dev_ptr(t, dtype, numel):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype and t.numel() == numel):
        raise _lib.SrxError(f"expected a contiguous device tensor of {numel} x {dtype}")
    return C.c_void_p(t.data_ptr())


def post_process_raw(accum4, res_x, res_y, tonemap=TONEMAP_NONE, want_float=False):
    """Returns the RGBA8 canvas (device uint8 tensor [res_y, res_x, 4]) and, if asked, the float RGB in [0, 255]
    that `post_process_pixel` returns (device [res_y, res_x, 3])."""
    import torch
    n = int(res_x) * int(res_y)
    a = _dev_ptr(accum4, torch.float32, 4 * n)
    rgba = torch.empty((res_y, res_x, 4), dtype=torch.uint8, device=accum4.device)
    rgb = torch.empty((res_y, res_x, 3), dtype=torch.float32, device=accum4.device) if want_float else None
    torch.cuda.synchronize(accum4.device)
    with torch.cuda.device(accum4.device):
        check(_lib.lib().srx_post_process(a, C.c_int32(res_x), C.c_int32(res_y), C.c_int32(tonemap), C.c_void_p(rgba.data_ptr()),
                                          C.c_void_p(rgb.data_ptr()) if want_float else None), "srx_post_process")
    return (rgba, rgb) if want_float else rgba


def write_raw_file(accum4, res_x, res_y, path):
    import torch
    a = _dev_ptr(accum4, torch.float32, 4 * int(res_x) * int(res_y))
    torch.cuda.synchronize(accum4.device)
    with torch.cuda.device(accum4.device):
        check(_lib.lib().srx_write_raw(str(path).encode(), a, C.c_int32(res_x), C.c_int32(res_y)), "srx_write_raw")


def read_raw(path):
    """(rgb[res_y, res_x, 3] float32 host array, res_x, res_y); needs no GPU."""
    L = _lib.lib()
    rx, ry = C.c_int32(0), C.c_int32(0)
    check(L.srx_read_raw(str(path).encode(), None, C.c_uint64(0), C.byref(rx), C.byref(ry)), "srx_read_raw")
    out = np.zeros((ry.value, rx.value, 3), np.float32)
    check(L.srx_read_raw(str(path).encode(), out.ctypes.data_as(C.c_void_p), C.c_uint64(rx.value * ry.value), C.byref(rx),
                         C.byref(ry)), "srx_read_raw")
    return out, rx.value, ry.value


def write_png(rgba8, path):
    import torch
    res_y, res_x = int(rgba8.shape[0]), int(rgba8.shape[1])
    p = _dev_ptr(rgba8, torch.uint8, 4 * res_x * res_y)
    torch.cuda.synchronize(rgba8.device)
    with torch.cuda.device(rgba8.device):
        check(_lib.lib().srx_write_png(str(path).encode(), p, C.c_int32(res_x), C.c_int32(res_y)), "srx_write_png")
