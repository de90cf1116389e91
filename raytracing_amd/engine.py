"""Host-side mirror of the reference's trace-and-shade interface over the srx C-ABI.

Names follow the reference (`spatialPartitioning/`): `PackagedScene` (Scene.h:18-31) built by
`package()` (Scene::package, Scene.h:90-179); `RayEngine` (Core.h:36-48) with `load_scene`,
`execute_ray_cast` (Core.h:346), `execute_lighting_cast` (Core.h:351) and `process_ray` (Core.h:359);
`SceneManager` (SceneManager.h:260) reduced to what the hot path needs: sample passes with the
running-mean accumulation of `RenderThread::process_iteratively` (SceneManager.h:193-237).
The calls are batched: one call casts/shades an array of rays on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SrxCounters, SrxSceneDesc, SrxTexture, check

HIT_DTYPE = np.dtype([("material", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri_index", "<i4")])
MISS = 0xFFFFFFFF


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dev_ptr(x):
    """Accept a torch CUDA tensor or a raw device address."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("framebuffers must be contiguous device tensors")
        # The library works on streams of its own, which do not order against torch's: whatever torch has queued for
        # this tensor (its zero fill, a priority update) must be over before the library touches it. The calls
        # themselves return only when their work is complete, so nothing is needed on the way back.
        import torch
        torch.cuda.current_stream(x.device).synchronize()
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class PackagedScene:
    """The flattened scene handed to the engine: three flat arrays in the reference's byte layouts
    (BVH_AVX 288 B, PTri_AVX 448 B, PackagedTri 196 B) plus materials, textures, lights, parameters."""

    def __init__(self, nodes, packets, tris, materials, textures=(), suns=(), max_generations=10,
                 monte_carlo_max=32, monte_carlo_modifier=0.25, tables=None):
        self.avx_bvh = np.ascontiguousarray(nodes, np.uint8).reshape(-1, 288)
        self.avx_tri_data = np.ascontiguousarray(packets, np.uint8).reshape(-1, 448)
        self.tri_data = np.ascontiguousarray(tris, np.uint8).reshape(-1, 196)
        self.materials = np.ascontiguousarray(materials)
        self.textures = list(textures)
        self.lights = list(suns)
        self.max_generations = int(max_generations)
        self.monte_carlo_max = int(monte_carlo_max)
        self.monte_carlo_modifier = float(monte_carlo_modifier)
        self.tables = tables        # optional (arccos2048, pdf1m); built by the library when None


def package(desc, tables=None, fast=False) -> PackagedScene:
    """`Scene::package()` (Scene.h:90-179) through `srx_package_create`; fast=True: the opt-in builder of a different tree
    (`srx_package_create_fast`: not the reference's triangle numbering). `.order[k]` = input triangle of tri_data[k]."""
    L = _lib.lib()
    tris = np.ascontiguousarray(desc.tris)
    pkg = C.c_void_p()
    if fast:
        check(L.srx_package_create_fast(_p(tris), C.c_uint32(len(tris)), C.byref(pkg)), "srx_package_create_fast")
    else:
        check(L.srx_package_create(_p(tris), C.c_uint32(len(tris)), C.byref(pkg)), "srx_package_create")
    try:
        pn, pp, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nn, npk, nt = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(L.srx_package_get(pkg, C.byref(pn), C.byref(nn), C.byref(pp), C.byref(npk), C.byref(pt), C.byref(nt)), "srx_package_get")

        def grab(ptr, n, sz):
            if n == 0:
                return np.zeros((0, sz), np.uint8)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * sz,)).copy().reshape(n, sz)

        nodes, packets, ptris = grab(pn, nn.value, 288), grab(pp, npk.value, 448), grab(pt, nt.value, 196)
        po, no = C.POINTER(C.c_uint32)(), C.c_uint32()
        check(L.srx_package_get_order(pkg, C.byref(po), C.byref(no)), "srx_package_get_order")
        order = np.ctypeslib.as_array(po, shape=(no.value,)).copy() if no.value else np.zeros(0, np.uint32)
    finally:
        L.srx_package_destroy(pkg)
    PS = PackagedScene(nodes, packets, ptris, desc.materials, desc.textures, desc.suns, desc.max_generations,
                       desc.monte_carlo_max, desc.monte_carlo_modifier, tables)
    PS.order = order
    return PS


def build_tables():
    """(arccos_lookup[2048], pdf_lookup[1024*1024]) — VecLib.cpp:391-398, PDF.h:83-123."""
    a = np.zeros(2048, np.float32)
    p = np.zeros(1024 * 1024, np.float32)
    check(_lib.lib().srx_build_tables(_p(a), _p(p)), "srx_build_tables")
    return a, p


class RayEngine:
    """`RayEngine` (Core.h:36-48) on the GPU."""

    def __init__(self, device=0):
        self.L = _lib.lib()
        check(self.L.srx_init(C.c_int(device)), "srx_init")
        self.scene = None
        self.data = None
        self._renderer = None
        self._renderer_capacity = 0

    def load_scene(self, PS: PackagedScene):
        """`RayEngine::load_scene(PackagedScene*)` (Core.h:46-48): deep-copies the scene to the device."""
        self.close()
        d = SrxSceneDesc()
        keep = []
        d.nodes, d.n_nodes = _p(PS.avx_bvh), len(PS.avx_bvh)
        d.packets, d.n_packets = _p(PS.avx_tri_data), len(PS.avx_tri_data)
        d.tris, d.n_tris = _p(PS.tri_data), len(PS.tri_data)
        d.materials, d.n_materials = _p(PS.materials), len(PS.materials)
        tex = (SrxTexture * max(1, len(PS.textures)))()
        for i, t in enumerate(PS.textures):
            img = np.ascontiguousarray(t.rgb8, np.uint8)
            keep.append(img)
            tex[i].rgb8 = img.ctypes.data
            tex[i].width, tex[i].height = img.shape[1], img.shape[0]
            tex[i].scale_x, tex[i].scale_y = t.scale
            tex[i].normal_transform = int(t.normal_transform)
        d.textures, d.n_textures = tex, len(PS.textures)
        suns = np.zeros((max(1, len(PS.lights)), 6), np.float32)
        for i, (dr, em) in enumerate(PS.lights):
            suns[i, :3] = dr
            suns[i, 3:] = em
        d.suns, d.n_suns = _p(suns), len(PS.lights)
        d.max_generations, d.monte_carlo_max, d.monte_carlo_modifier = PS.max_generations, PS.monte_carlo_max, PS.monte_carlo_modifier
        if PS.tables is not None:
            a = np.ascontiguousarray(PS.tables[0], np.float32)
            p = np.ascontiguousarray(PS.tables[1], np.float32)
            keep += [a, p]
            d.arccos_lookup, d.pdf_lookup = _p(a), _p(p)
        h = C.c_void_p()
        check(self.L.srx_scene_create(C.byref(d), C.byref(h)), "srx_scene_create")
        self.scene = h
        self.data = PS

    def close(self):
        if self._renderer is not None:
            self.L.srx_renderer_destroy(self._renderer)
            self._renderer = None
        if self.scene is not None:
            self.L.srx_scene_destroy(self.scene)
            self.scene = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stack_bound(self):
        return self.L.srx_scene_stack_bound(self.scene)

    def stack_spills(self):
        """How often a traversal wave has moved part of its stack to the scene's overflow areas (deep trees only)."""
        return int(self.L.srx_scene_stack_spills(self.scene))

    # ---- casts ---------------------------------------------------------
    def execute_ray_cast(self, position, slope, want_advanced=False):
        """`execute_ray_cast(XYZ& position, const XYZ& slope)` (Core.h:346) over a batch.

        Returns (hits, counters[, advanced positions]); counters = (node tests, packet tests)."""
        org = np.ascontiguousarray(position, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(slope, np.float32).reshape(-1, 3)
        if org.shape != dirs.shape:
            raise ValueError("position and slope must have the same shape")
        n = len(org)
        hits = np.zeros(n, HIT_DTYPE)
        adv = np.zeros((n, 3), np.float32) if want_advanced else None
        ctr = np.zeros(2, np.uint64)
        check(self.L.srx_cast_closest(self.scene, _p(org), _p(dirs), C.c_uint32(n), _p(hits), _p(adv), _p(ctr)), "srx_cast_closest")
        return (hits, ctr, adv) if want_advanced else (hits, ctr)

    def execute_lighting_cast(self, position, light_direction):
        """`execute_lighting_cast(XYZ position, Light*)` (Core.h:351-358) for sun lights over a batch."""
        org = np.ascontiguousarray(position, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(np.broadcast_to(np.asarray(light_direction, np.float32), org.shape))
        lit = np.zeros(len(org), np.uint8)
        check(self.L.srx_cast_shadow(self.scene, _p(org), _p(dirs), C.c_uint32(len(org)), _p(lit)), "srx_cast_shadow")
        return lit

    # ---- shading -------------------------------------------------------
    def renderer(self, ray_capacity=1 << 22):
        if self._renderer is None or self._renderer_capacity < ray_capacity:
            if self._renderer is not None:
                self.L.srx_renderer_destroy(self._renderer)
            r = C.c_void_p()
            check(self.L.srx_renderer_create(self.scene, C.c_uint64(ray_capacity), C.byref(r)), "srx_renderer_create")
            self._renderer, self._renderer_capacity = r, ray_capacity
        return self._renderer

    def set_timing(self, enabled=True, ray_capacity=1 << 22):
        check(self.L.srx_renderer_set_timing(self.renderer(ray_capacity), C.c_int(int(enabled))), "srx_renderer_set_timing")

    def process_ray(self, position, slope, keys, generation=0, ray_capacity=1 << 22):
        """`process_ray(Casting_Diagnostics&, PackagedRay&)` (Core.h:359-601) over a batch of rays.

        Ray i draws its random numbers from the keyed stream `keys[i]`. Returns (radiance[n,3], counters)."""
        org = np.ascontiguousarray(position, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(slope, np.float32).reshape(-1, 3)
        keys = np.ascontiguousarray(keys, np.uint64)
        rad = np.zeros((len(org), 3), np.float32)
        ctr = SrxCounters()
        check(self.L.srx_process_rays(self.renderer(ray_capacity), _p(org), _p(dirs), _p(keys), C.c_int32(generation),
                                      C.c_uint32(len(org)), _p(rad), C.byref(ctr)), "srx_process_rays")
        return rad, ctr.as_dict()

    def trace_visit(self, camera, rect, visit_index, seed, accum4, aux4=None, block_priority=None, ray_capacity=1 << 22,
                    interleave=None, deal=None, part=0):
        """One visit of every 4x4 block (`RenderThread::process_iteratively`, SceneManager.h:193-237) with the
        controller's block priorities (device tensor, None = all 1): a block takes a sample with probability
        `priority`; a pixel's sample index is its own count so far."""
        cam = np.ascontiguousarray(camera)
        ctr = SrxCounters()
        if deal is not None:
            # the whole frame's rows as `deal` (tiles.Deal) gives them to `part`: srx_trace_pass_adaptive_dealt
            check(self.L.srx_trace_pass_adaptive_dealt(self.renderer(ray_capacity), _p(cam), deal.h, C.c_int32(part), C.c_uint32(visit_index),
                                                       C.c_uint64(seed), _dev_ptr(block_priority), _dev_ptr(accum4), _dev_ptr(aux4),
                                                       C.byref(ctr)), "srx_trace_pass_adaptive_dealt")
            return ctr.as_dict()
        x0, y0, w, h = rect
        group, part, nparts = interleave if interleave is not None else (max(1, h), 0, 1)
        check(self.L.srx_trace_pass_adaptive(self.renderer(ray_capacity), _p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w),
                                             C.c_int32(h), C.c_int32(group), C.c_int32(part), C.c_int32(nparts),
                                             C.c_uint32(visit_index), C.c_uint64(seed), _dev_ptr(block_priority), _dev_ptr(accum4),
                                             _dev_ptr(aux4), C.byref(ctr)), "srx_trace_pass_adaptive")
        return ctr.as_dict()

    def trace_pass(self, camera, rect, pass_index, seed, accum4, aux4=None, ray_capacity=1 << 22, interleave=None, n_passes=1):
        """`n_passes` consecutive sample passes (default one) of `RenderThread::process_iteratively` (SceneManager.h:193-237)
        over a pixel rectangle; accum4/aux4 are device float4 framebuffers (torch tensors or raw addresses). Several
        passes in one call share their kernel launches and give the same samples as one call per pass.

        `interleave=(group_rows, part, nparts)` renders only this caller's share of the rows: row groups
        part, part + nparts, ... (the reference's interleaved block assignment, SceneManager.h:441-459)."""
        cam = np.ascontiguousarray(camera)
        x0, y0, w, h = rect
        group, part, nparts = interleave if interleave is not None else (max(1, h), 0, 1)
        ctr = SrxCounters()
        check(self.L.srx_trace_passes_interleaved(self.renderer(ray_capacity), _p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w),
                                                  C.c_int32(h), C.c_int32(group), C.c_int32(part), C.c_int32(nparts),
                                                  C.c_uint32(pass_index), C.c_int32(n_passes), C.c_uint64(seed), _dev_ptr(accum4),
                                                  _dev_ptr(aux4), C.byref(ctr)), "srx_trace_passes_interleaved")
        return ctr.as_dict()


def camera_rays(camera, rect, pass_index, seed):
    """`Camera::random_slope_at` (Camera.h:138-140) for a pixel rectangle: (directions[n,3], keys[n])."""
    cam = np.ascontiguousarray(camera)
    x0, y0, w, h = rect
    d = np.zeros((w * h, 3), np.float32)
    k = np.zeros(w * h, np.uint64)
    check(_lib.lib().srx_camera_rays(_p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w), C.c_int32(h),
                                     C.c_uint32(pass_index), C.c_uint64(seed), _p(d), _p(k)), "srx_camera_rays")
    return d, k


def camera_rays_xorgen(camera, rect, pass_index, seed4=(1, 2, 3, 4)):
    """`Camera::random_slope_at` with the reference's own `XorGen` (XorRandom.cpp), one sequential stream seeded with
    `seed4`, two draws per pixel in row-major order: the parity ray sets of fixed XorRandom seeds."""
    cam = np.ascontiguousarray(camera)
    x0, y0, w, h = rect
    d = np.zeros((w * h, 3), np.float32)
    s4 = np.asarray(seed4, np.uint64)
    check(_lib.lib().srx_camera_rays_xorgen(_p(cam), C.c_int32(x0), C.c_int32(y0), C.c_int32(w), C.c_int32(h),
                                            C.c_uint32(pass_index), _p(s4), _p(d)), "srx_camera_rays_xorgen")
    return d


def xorgen_stream(seed4, n_pools, n_scalar):
    """(pools[n_pools, 8], scalars[n_scalar]) of one `XorGen` seeded with seed4, computed on the device."""
    s4 = np.asarray(seed4, np.uint64)
    po = np.zeros((n_pools, 8), np.float32)
    sc = np.zeros(n_scalar, np.float32)
    check(_lib.lib().srx_xorgen_stream(_p(s4), C.c_uint32(n_pools), C.c_uint32(n_scalar), _p(po), _p(sc)), "srx_xorgen_stream")
    return po, sc


REGION_DTYPE = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("active", "<i4"), ("noise", "<f4"),
                         ("max_noise", "<f4"), ("min_noise", "<f4"), ("avg_noise", "<f4")])


class NoiseController:
    """Adaptive-sampling controller: `SceneManager::analyze_noise` (SceneManager.h:526-667) and the priority half of
    `SceneManager::balance_threads` (:668-692). Framebuffers, error map and priorities are device tensors."""

    BLOCK = 4                                 # SceneManager.h:299

    ABS_LUMINANCE = 1                         # SRX_NOISE_ABS_LUMINANCE

    def __init__(self, res_x, res_y, seed=0, minimum_samples=-1, device=0, abs_luminance=False):
        self.L = _lib.lib()
        check(self.L.srx_init(C.c_int(device)), "srx_init")
        self.res_x, self.res_y = int(res_x), int(res_y)
        self.h = C.c_void_p()
        check(self.L.srx_noise_create(C.c_int32(res_x), C.c_int32(res_y), C.c_uint64(seed), C.c_int32(minimum_samples),
                                      C.byref(self.h)), "srx_noise_create")
        if abs_luminance:
            # opt-in, not the reference: negative luminances do not turn the whole controller NaN (quirk Q11)
            check(self.L.srx_noise_set_option(self.h, C.c_int32(self.ABS_LUMINANCE), C.c_int32(1)), "srx_noise_set_option")

    def close(self):
        if self.h:
            self.L.srx_noise_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_blocks(self):
        return ((self.res_x + 3) // 4) * ((self.res_y + 3) // 4)

    def error_map(self, accum4, aux4, err, interleave=None, deal=None, part=0):
        """Fills `err` (device, one float per pixel) for this caller's rows, 0 elsewhere; returns the minimum sample
        count of those rows (SceneManager.h:492-501, :566-572). `deal` (tiles.Deal) + `part`: the rows a deal gives it."""
        mn = C.c_int32(0)
        if deal is not None:
            check(self.L.srx_noise_error_map_dealt(self.h, _dev_ptr(accum4), _dev_ptr(aux4), deal.h, C.c_int32(part), _dev_ptr(err),
                                                   C.byref(mn)), "srx_noise_error_map_dealt")
            return mn.value
        group, part, nparts = interleave if interleave is not None else (self.res_y, 0, 1)
        check(self.L.srx_noise_error_map(self.h, _dev_ptr(accum4), _dev_ptr(aux4), C.c_int32(group), C.c_int32(part),
                                         C.c_int32(nparts), _dev_ptr(err), C.byref(mn)), "srx_noise_error_map")
        return mn.value

    def analyze_noise(self, err, seen):
        check(self.L.srx_noise_analyze(self.h, _dev_ptr(err), C.c_int32(seen)), "srx_noise_analyze")

    def regions(self):
        n = C.c_uint32(0)
        check(self.L.srx_noise_regions(self.h, None, C.c_uint32(0), C.byref(n)), "srx_noise_regions")
        out = np.zeros(n.value, REGION_DTYPE)
        check(self.L.srx_noise_regions(self.h, _p(out), C.c_uint32(n.value), C.byref(n)), "srx_noise_regions")
        return out

    def balance(self, seen, block_priority):
        """Writes the block priorities (device, `n_blocks` floats); False while `seen` < minimum_samples (all 1)."""
        upd = C.c_int32(0)
        check(self.L.srx_noise_priorities(self.h, C.c_int32(seen), _dev_ptr(block_priority), C.byref(upd)), "srx_noise_priorities")
        return bool(upd.value)
