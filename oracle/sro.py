"""ctypes binding of the CPU oracle (oracle/libsro.so) — TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (raytracing_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIT_DTYPE = np.dtype([("material", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri_index", "<i4")])


def build(force=False):
    so = os.path.join(_HERE, "libsro.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "libsro.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsro.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.sro_last_error.restype = C.c_char_p
        L.sro_scene_new.restype = C.c_void_p
        for name in ("sro_scene_nodes", "sro_scene_packets", "sro_scene_tris", "sro_scene_arccos", "sro_scene_pdf"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.sro_scene_build_seconds.restype = C.c_double
        L.sro_scene_build_seconds.argtypes = [C.c_void_p]
        L.sro_primary_key.restype = C.c_uint64
        L.sro_primary_key.argtypes = [C.c_uint64, C.c_uint64]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """CPU restatement of PackagedScene + RayEngine for one scene description."""

    def __init__(self, desc=None, build_tables=True, tables=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.sro_scene_new())
        self._keep = []
        if desc is not None:
            self.load(desc, build_tables=build_tables, tables=tables)

    def __del__(self):
        try:
            if self.h:
                self.L.sro_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.L.sro_last_error().decode())

    def load(self, desc, build_tables=True, tables=None):
        tris = np.ascontiguousarray(desc.tris)
        self._chk(self.L.sro_scene_package(self.h, _p(tris), C.c_uint32(len(tris))))
        mats = np.ascontiguousarray(desc.materials)
        self._chk(self.L.sro_scene_set_materials(self.h, _p(mats), C.c_uint32(len(mats))))
        for t in desc.textures:
            img = np.ascontiguousarray(t.rgb8)
            self.L.sro_scene_add_texture(self.h, _p(img), C.c_int32(img.shape[1]), C.c_int32(img.shape[0]),
                                         C.c_float(t.scale[0]), C.c_float(t.scale[1]), C.c_int32(int(t.normal_transform)))
        for d, e in desc.suns:
            dd = np.asarray(d, np.float32); ee = np.asarray(e, np.float32)
            self.L.sro_scene_add_sun(self.h, _p(dd), _p(ee))
        self.L.sro_scene_set_params(self.h, C.c_int32(desc.max_generations), C.c_int32(desc.monte_carlo_max),
                                    C.c_float(desc.monte_carlo_modifier))
        if tables is not None:
            a, p = tables
            self.L.sro_scene_set_tables(self.h, _p(np.ascontiguousarray(a, np.float32)), _p(np.ascontiguousarray(p, np.float32)))
        elif build_tables:
            self.L.sro_scene_build_tables(self.h)

    def set_leaf_filter(self, enabled):
        """Mirror of the product's SRX_LEAF_FILTER switch: off = count no packet test as saved."""
        self.L.sro_scene_set_leaf_filter(self.h, C.c_int32(int(bool(enabled))))

    def set_simd(self, enabled):
        """Casts through the AVX2-intrinsic node / packet tests (oracle/sro_trace_avx2.h): bit-identical results, no
        leaf-filter bookkeeping; the timed CPU baseline of bench.py."""
        self.L.sro_scene_set_simd(self.h, C.c_int32(int(bool(enabled))))

    def set_params(self, max_generations, monte_carlo_max, monte_carlo_modifier):
        self.L.sro_scene_set_params(self.h, C.c_int32(max_generations), C.c_int32(monte_carlo_max), C.c_float(monte_carlo_modifier))

    # -- packaged arrays in the reference layouts (288 / 448 / 196 bytes per record)
    def counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.sro_scene_counts(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def packaged(self):
        nn, npk, nt = self.counts()
        def grab(ptr, n, sz):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * sz,)).copy().reshape(n, sz)
        return (grab(self.L.sro_scene_nodes(self.h), nn, 288), grab(self.L.sro_scene_packets(self.h), npk, 448),
                grab(self.L.sro_scene_tris(self.h), nt, 196))

    def tables(self):
        a = np.ctypeslib.as_array(C.cast(self.L.sro_scene_arccos(self.h), C.POINTER(C.c_float)), shape=(2048,)).copy()
        p = np.ctypeslib.as_array(C.cast(self.L.sro_scene_pdf(self.h), C.POINTER(C.c_float)), shape=(1024 * 1024,)).copy()
        return a, p

    def build_seconds(self):
        return self.L.sro_scene_build_seconds(self.h)

    # -- casts
    def cast_closest(self, org, dirs, want_advanced=False):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(org)
        out = np.zeros(n, HIT_DTYPE)
        adv = np.zeros((n, 3), np.float32) if want_advanced else None
        ctr = np.zeros(7, np.uint64)
        self._chk(self.L.sro_cast_closest(self.h, _p(org), _p(dirs), C.c_uint32(n), _p(out), _p(adv), _p(ctr)))
        return (out, ctr, adv) if want_advanced else (out, ctr)

    def filter_margins(self, org, dirs):
        """(largest |ray point - triangle point| / |o - p1|, largest distance outside the leaf's tight box / D, lanes) over
        every lane the reference's triangle test passes in a leaf whose guard condition holds for the ray (sro_trace.h)."""
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros(3, np.float64)
        self._chk(self.L.sro_filter_margins(self.h, _p(org), _p(dirs), C.c_uint32(len(org)), _p(out)))
        return float(out[0]), float(out[1]), int(out[2])

    def cast_shadow(self, org, dirs):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        lit = np.zeros(len(org), np.uint8)
        self._chk(self.L.sro_cast_shadow(self.h, _p(org), _p(dirs), C.c_uint32(len(org)), _p(lit)))
        return lit

    def process_rays_keyed(self, org, dirs, keys, generation=0):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        keys = np.ascontiguousarray(keys, np.uint64)
        rad = np.zeros((len(org), 3), np.float32)
        st = np.zeros(10, np.uint64)
        self._chk(self.L.sro_process_rays_keyed(self.h, _p(org), _p(dirs), _p(keys), C.c_int32(generation),
                                                C.c_uint32(len(org)), _p(rad), _p(st)))
        return rad, st

    def render_pass(self, camera, rect, pass_index, seed, accum4, aux4=None, mode=0, threads=1):
        cam = np.ascontiguousarray(camera)
        x0, y0, w, h = rect
        st = np.zeros(10, np.uint64)
        self._chk(self.L.sro_render_pass(self.h, _p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w), C.c_int32(h),
                                         C.c_uint32(pass_index), C.c_uint64(seed), C.c_int32(mode), C.c_int32(threads),
                                         _p(accum4), _p(aux4), _p(st)))
        return st


def _render_pass_adaptive(self, camera, rect, visit_index, seed, accum4, aux4=None, block_priority=None, threads=1):
    """One visit of every 4x4 block with per-block priorities (SceneManager.h:193-237), keyed RNG."""
    cam = np.ascontiguousarray(camera)
    x0, y0, w, h = rect
    st = np.zeros(10, np.uint64)
    pr = None if block_priority is None else np.ascontiguousarray(block_priority, np.float32)
    self._chk(self.L.sro_render_pass_adaptive(self.h, _p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w), C.c_int32(h),
                                              C.c_uint32(visit_index), C.c_uint64(seed), _p(pr), C.c_int32(threads),
                                              _p(accum4), _p(aux4), _p(st)))
    return st


OracleScene.render_pass_adaptive = _render_pass_adaptive


def camera_rays_keyed(camera, rect, pass_index, seed):
    cam = np.ascontiguousarray(camera)
    x0, y0, w, h = rect
    d = np.zeros((w * h, 3), np.float32)
    k = np.zeros(w * h, np.uint64)
    lib().sro_camera_rays_keyed(_p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w), C.c_int32(h), C.c_uint32(pass_index),
                                C.c_uint64(seed), _p(d), _p(k))
    return d, k


def camera_rays_xorgen(camera, rect, pass_index, seed4=(1, 2, 3, 4)):
    """Primaries of a rectangle, row-major, jitter from ONE sequential XorGen seeded with seed4."""
    cam = np.ascontiguousarray(camera)
    x0, y0, w, h = rect
    d = np.zeros((w * h, 3), np.float32)
    s4 = np.asarray(seed4, np.uint64)
    lib().sro_camera_rays_xorgen(_p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w), C.c_int32(h), C.c_uint32(pass_index), _p(s4), _p(d))
    return d


def first_bounce_rays(S, org, dirs, seed4=(5, 6, 7, 8), capacity=None):
    """Child rays queued by generation-0 `process_ray` calls in reference mode (one sequential XorGen seeded with
    seed4, Core.h:498-591, :573-588): (origins, directions, parent index, total queued)."""
    org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    cap = int(capacity if capacity is not None else 40 * len(org))
    co = np.zeros((cap, 3), np.float32); cd = np.zeros((cap, 3), np.float32); par = np.zeros(cap, np.uint32)
    n = C.c_uint32(0)
    s4 = np.asarray(seed4, np.uint64)
    S._chk(S.L.sro_first_bounce_rays(S.h, _p(org), _p(dirs), C.c_uint32(len(org)), _p(s4), C.c_uint32(cap), _p(co), _p(cd), _p(par), C.byref(n)))
    m = min(cap, n.value)
    return co[:m], cd[:m], par[:m], n.value


def xorgen_stream(seed4, n_scalar, n_pools):
    s = np.asarray(seed4, np.uint64)
    sc = np.zeros(n_scalar, np.float32); po = np.zeros((n_pools, 8), np.float32); raw = np.zeros(4, np.uint64)
    lib().sro_xorgen_stream(_p(s), C.c_uint32(n_scalar), C.c_uint32(n_pools), _p(sc), _p(po), _p(raw))
    return sc, po, raw


def keyed_stream(key, n_pools, n_scalar):
    po = np.zeros((n_pools, 8), np.float32); sc = np.zeros(n_scalar, np.float32); ck = C.c_uint64()
    lib().sro_keyed_stream(C.c_uint64(key), C.c_uint32(n_pools), C.c_uint32(n_scalar), _p(po), _p(sc), C.byref(ck))
    return po, sc, ck.value


def trig(x):
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x); t = np.zeros_like(x)
    lib().sro_trig(_p(x), C.c_uint32(len(x)), _p(s), _p(c), _p(t))
    return s, c, t


def avx_ops(fields):
    """fields[m, 10, 8] (a.xyz, b.xyz, c.xyz, s per lane) -> [m, 5, 8]: cross(a, b).xyz, dot(a, b), s * dot(a, c)
    with the FMA nesting of VecLib.cpp:282-325 (what the packet test is made of)."""
    f = np.ascontiguousarray(fields, np.float32).reshape(-1, 10, 8)
    out = np.zeros((len(f), 5, 8), np.float32)
    lib().sro_avx_ops(_p(f), C.c_uint32(len(f)), _p(out))
    return out


def poly_acos(x):
    x = np.ascontiguousarray(x, np.float32)
    o = np.zeros_like(x)
    lib().sro_poly_acos(_p(x), C.c_uint32(len(x)), _p(o))
    return o


# ---- adaptive-sampling controller (oracle/sro_noise.h) ------------------------------------------------------
def noise_error_map(accum4, aux4, abs_luminance=False):
    """(err[npix], minimum sample count): SceneManager.h:566-572 per pixel, :492-501. abs_luminance mirrors the product's
    opt-in SRX_NOISE_ABS_LUMINANCE (not the reference)."""
    a = np.ascontiguousarray(accum4, np.float32).reshape(-1, 4)
    x = np.ascontiguousarray(aux4, np.float32).reshape(-1, 4)
    err = np.zeros(len(a), np.float32)
    mn = C.c_int32(0)
    lib().sro_noise_error_map(_p(a), _p(x), C.c_uint64(len(a)), _p(err), C.byref(mn), C.c_int32(int(bool(abs_luminance))))
    return err, mn.value


def noise_block_fires(priority, seed, visit_index, block, n_blocks):
    return bool(lib().sro_noise_block_fires(C.c_float(priority), C.c_uint64(seed), C.c_uint32(visit_index), C.c_uint32(block),
                                            C.c_uint32(n_blocks)))


class OracleNoise:
    """SceneManager::analyze_noise / balance_threads (SceneManager.h:526-692) on the CPU."""

    def __init__(self, res_x, res_y, seed, minimum_samples=64):
        L = lib()
        L.sro_noise_new.restype = C.c_void_p
        L.sro_noise_region_count.restype = C.c_uint32
        self.L, self.res_x, self.res_y = L, res_x, res_y
        self.h = C.c_void_p(L.sro_noise_new(C.c_int32(res_x), C.c_int32(res_y), C.c_uint64(seed), C.c_int32(minimum_samples)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sro_noise_free(self.h)
            self.h = None

    def analyze(self, err, seen):
        e = np.ascontiguousarray(err, np.float32)
        assert e.size == self.res_x * self.res_y
        self.L.sro_noise_analyze(self.h, _p(e), C.c_int32(seen))

    def regions(self):
        n = self.L.sro_noise_region_count(self.h)
        rects = np.zeros((n, 4), np.int32)
        stats = np.zeros((n, 5), np.float32)
        self.L.sro_noise_regions(self.h, _p(rects), _p(stats))
        return rects, stats

    def priorities(self, seen):
        bx, by = (self.res_x + 3) // 4, (self.res_y + 3) // 4
        out = np.ones(bx * by, np.float32)
        wrote = self.L.sro_noise_priorities(self.h, C.c_int32(seen), _p(out))
        return out, bool(wrote)


# ---- output stage (oracle/sro_post.h) -----------------------------------------------------------------------
def post_process(rgb, tonemap=0):
    """ImageHandler::post_process_pixel over an array of linear RGB pixels: (float out in [0, 255], RGBA8)."""
    a = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros_like(a)
    u8 = np.zeros((len(a), 4), np.uint8)
    lib().sro_post_process(_p(a), C.c_uint64(len(a)), C.c_int(tonemap), _p(out), _p(u8))
    return out, u8


def write_raw(path, rgb, res_x, res_y):
    """FileManager::writeRaw (FileLoading.h:171-182): u32 resX, u32 resY, then rows of float3 pixels, row 0 first."""
    a = np.ascontiguousarray(rgb, np.float32).reshape(res_y, res_x, 3)
    with open(path, "wb") as fh:
        fh.write(np.array([res_x, res_y], np.uint32).tobytes())
        fh.write(a.tobytes())


def read_raw(path):
    """FileManager::readRaw (FileLoading.h:183-199): returns (rgb[res_y, res_x, 3], res_x, res_y)."""
    with open(path, "rb") as fh:
        rx, ry = (int(v) for v in np.frombuffer(fh.read(8), np.int32))
        a = np.frombuffer(fh.read(rx * ry * 12), np.float32).reshape(ry, rx, 3).copy()
    return a, rx, ry
