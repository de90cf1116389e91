"""Scene ingestion: host-side mirror of `FileManager::loadGLTFFile` (FileLoading.h:200-224) over the
C-ABI (`srx_gltf_load`, csrc/srx_gltf.cpp): a binary glTF becomes a `scenes.SceneDesc` — world-space triangles in the
order `Scene::package` collects them, material table, textures as the reference stores them, suns, first camera — ready
for `engine.package` / `RayEngine.load_scene`. Host-only: needs no GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib, scenes
from ._lib import check, SrxTexture

SUN_DTYPE = np.dtype([("direction", "<f4", (3,)), ("emission", "<f4", (3,))])


def load_gltf_file(path, res_x, res_y, subdiv=1, max_generations=10, monte_carlo_max=32, monte_carlo_modifier=0.25):
    """`loadGLTFFile` + the render settings `main` adds (Main.cpp:49-205: resolution, subdivision; Scene.h:27-30 defaults)."""
    L = _lib.lib()
    h = C.c_void_p()
    check(L.srx_gltf_load(os.fspath(path).encode(), C.byref(h)), "srx_gltf_load")
    try:
        tris, mats, tex, suns = C.c_void_p(), C.c_void_p(), C.POINTER(SrxTexture)(), C.c_void_p()
        nt, nm, nx, ns = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        cam = np.zeros((), scenes.CAMERA_DTYPE)
        check(L.srx_gltf_get(h, C.byref(tris), C.byref(nt), C.byref(mats), C.byref(nm), C.byref(tex), C.byref(nx), C.byref(suns),
                             C.byref(ns), cam.ctypes.data_as(C.c_void_p)), "srx_gltf_get")

        def grab(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            return np.frombuffer(C.string_at(ptr, n * dtype.itemsize), dtype).copy()

        desc = scenes.SceneDesc(name=os.path.basename(os.fspath(path)), tris=grab(tris, nt.value, scenes.TRI_DTYPE),
                                materials=grab(mats, nm.value, scenes.MATERIAL_DTYPE))
        for i in range(nx.value):
            t = tex[i]
            rgb = np.frombuffer(C.string_at(t.rgb8, t.width * t.height * 3), np.uint8).reshape(t.height, t.width, 3).copy()
            desc.textures.append(scenes.TextureDesc(rgb, (t.scale_x, t.scale_y), bool(t.normal_transform)))
        for s in grab(suns, ns.value, SUN_DTYPE):
            desc.suns.append((s["direction"].copy(), s["emission"].copy()))
    finally:
        L.srx_gltf_destroy(h)
    cam["res_x"] = res_x; cam["res_y"] = res_y; cam["subdiv"] = subdiv
    desc.camera = cam
    desc.max_generations = max_generations
    desc.monte_carlo_max = monte_carlo_max
    desc.monte_carlo_modifier = monte_carlo_modifier
    return desc


def decode_image(data):
    """Embedded image bytes (8-bit PNG or baseline JPEG) -> (h, w, 4) uint8, as `stbi_load_from_memory(..., 4)` gives
    them to tinygltf's LoadImageData (`srx_image_decode`)."""
    L = _lib.lib()
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    check(L.srx_image_decode(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(buf)), C.byref(w), C.byref(h), None, C.c_size_t(0)), "srx_image_decode")
    out = np.zeros((h.value, w.value, 4), np.uint8)
    check(L.srx_image_decode(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(buf)), C.byref(w), C.byref(h),
                             out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)), "srx_image_decode")
    return out
