"""ctypes loader for raytracing_amd/libsrx.so (the C-ABI of include/srx.h).

There is no fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRX_LIB", os.path.join(HERE, "libsrx.so"))

# every symbol include/srx.h declares; tests check the library exports all of them
EXPORTS = [
    "srx_last_error", "srx_init",
    "srx_package_create", "srx_package_create_fast", "srx_package_get", "srx_package_get_order", "srx_package_destroy", "srx_build_tables",
    "srx_scene_create", "srx_scene_destroy", "srx_scene_stack_bound", "srx_scene_stack_spills",
    "srx_cast_closest", "srx_cast_closest_device", "srx_cast_closest_device_static", "srx_static_scratch_status", "srx_cast_shadow", "srx_cast_shadow_device",
    "srx_renderer_create", "srx_renderer_destroy", "srx_renderer_set_timing",
    "srx_trace_pass", "srx_trace_pass_interleaved", "srx_trace_passes_interleaved", "srx_trace_pass_adaptive", "srx_process_rays", "srx_camera_rays", "srx_camera_rays_xorgen", "srx_xorgen_stream",
    "srx_noise_create", "srx_noise_destroy", "srx_noise_set_option", "srx_noise_error_map", "srx_noise_analyze", "srx_noise_regions", "srx_noise_priorities",
    "srx_buffer_create", "srx_buffer_destroy", "srx_buffer_read", "srx_buffer_write", "srx_buffer_clear", "srx_post_process", "srx_write_raw", "srx_read_raw", "srx_write_png",
    "srx_gltf_load", "srx_gltf_destroy", "srx_gltf_get", "srx_image_decode", "srx_trace_dealing", "srx_trace_xcd_map",
    "srx_math_probe", "srx_deal_create", "srx_deal_destroy", "srx_deal_rebalance", "srx_deal_get", "srx_trace_pass_adaptive_dealt", "srx_noise_error_map_dealt", "srx_deal_slot_floats", "srx_rows_pack_dealt", "srx_rows_unpack_dealt", "srx_group_gather_framebuffer_dealt", "srx_group_allgather_rows_dealt", "srx_group_unique_id", "srx_group_create", "srx_group_destroy", "srx_group_rank", "srx_group_gather_framebuffer",
    "srx_group_allgather_rows", "srx_group_allreduce", "srx_rows_slot_floats", "srx_rows_pack", "srx_rows_unpack",
]


class SrxError(RuntimeError):
    pass


class SrxTexture(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("scale_x", C.c_float),
                ("scale_y", C.c_float), ("normal_transform", C.c_int32)]


class SrxSceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint32),
        ("packets", C.c_void_p), ("n_packets", C.c_uint32),
        ("tris", C.c_void_p), ("n_tris", C.c_uint32),
        ("materials", C.c_void_p), ("n_materials", C.c_uint32),
        ("textures", C.POINTER(SrxTexture)), ("n_textures", C.c_uint32),
        ("suns", C.c_void_p), ("n_suns", C.c_uint32),
        ("max_generations", C.c_int32), ("monte_carlo_max", C.c_int32), ("monte_carlo_modifier", C.c_float),
        ("pdf_lookup", C.c_void_p), ("arccos_lookup", C.c_void_p),
    ]


class SrxCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_processed", "diffuses_cast", "shadows_cast", "nodes", "packets",
                                           "shadow_nodes", "shadow_packets", "hits", "texels", "pdf_lookups",
                                           "trace_launches")] + [("trace_ms", C.c_double), ("shadow_launches", C.c_uint64),
                                                                 ("shadow_ms", C.c_double), ("shade_launches", C.c_uint64),
                                                                 ("shade_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_LIB = None


def lib():
    """Load libsrx.so; raises SrxError if it has not been built (run `python -m raytracing_amd.build`)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SrxError(f"{LIB_PATH} is missing: build it with `python -m raytracing_amd.build` (no CPU fallback exists)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; loading it first makes
        # libsrx.so bind to the same runtime, so device pointers and streams can be shared with torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.srx_last_error.restype = C.c_char_p
        L.srx_scene_stack_spills.restype = C.c_int64
        L.srx_rows_slot_floats.restype = C.c_uint64
        L.srx_deal_slot_floats.restype = C.c_uint64
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc != 0:
        raise SrxError(f"{what}: {lib().srx_last_error().decode()}")
