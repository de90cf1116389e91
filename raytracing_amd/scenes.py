"""Procedural scene descriptions for the trace-and-shade path.

The reference ships no scene assets (its `Assets/` holds UI icons only), so every
benchmark workload is generated here as plain data (`.glb` files go through `gltf.py`): world-space authoring triangles (what
`Object::fetchData`, `Primitives.h:423-434`, hands to `Scene::package`),
a material table (`Materials.h:317-358`), RGB8 textures (`Texture.h`), sun
lights (`FileLoading.h:724-732`) and a camera (`Camera.h`, `FileLoading.h:678-684`).

Conventions follow the reference: Y forward, Z up; the lens lies in the XZ
plane and `focal_position = (0, 0.5/tan(yfov/2), 0)`.

Quirk Q3: an exactly flat axis-aligned box is invisible to the
reference traversal, so tessellated planes get a >=1e-4 displacement.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# One authoring triangle: p1 p2 p3, UV_1..3, vertex normals, material, flags (bit0 = normals given).
TRI_DTYPE = np.dtype(
    [("p", "<f4", (9,)), ("uv", "<f4", (6,)), ("n", "<f4", (9,)), ("material", "<i4"), ("flags", "<u4")]
)
assert TRI_DTYPE.itemsize == 104

# Material table row; field order == `srx_material` in include/srx.h.
MATERIAL_DTYPE = np.dtype(
    [
        ("color", "<f4", (3,)), ("color_tex", "<i4"),
        ("roughness", "<f4"), ("roughness_tex", "<i4"),
        ("metallic", "<f4"), ("metallic_tex", "<i4"),
        ("specular", "<f4"), ("specular_tex", "<i4"),
        ("emissive", "<f4", (3,)), ("emissive_tex", "<i4"),
        ("normal", "<f4", (3,)), ("normal_tex", "<i4"),
        ("transmission", "<f4"), ("transmission_tex", "<i4"),
        ("ior", "<f4"), ("ior_tex", "<i4"),
        ("use_normals", "<i4"),
    ]
)
assert MATERIAL_DTYPE.itemsize == 92

CAMERA_DTYPE = np.dtype(
    [
        ("position", "<f4", (3,)), ("rotation", "<f4", (4,)), ("focal_position", "<f4", (3,)),
        ("lens_width", "<f4"), ("lens_height", "<f4"),
        ("res_x", "<i4"), ("res_y", "<i4"), ("subdiv", "<i4"),
    ]
)
assert CAMERA_DTYPE.itemsize == 60


@dataclass
class TextureDesc:
    rgb8: np.ndarray              # (h, w, 3) uint8, row-major
    scale: tuple = (1.0, 1.0)
    normal_transform: bool = False


@dataclass
class SceneDesc:
    name: str
    tris: np.ndarray                                   # TRI_DTYPE
    materials: np.ndarray                              # MATERIAL_DTYPE
    textures: list = field(default_factory=list)       # [TextureDesc]
    suns: list = field(default_factory=list)           # [(dir3, emission3)]
    max_generations: int = 10                          # Scene.h:27-30
    monte_carlo_max: int = 32
    monte_carlo_modifier: float = 0.25
    camera: np.ndarray | None = None                   # CAMERA_DTYPE scalar


def material(color=(0.8, 0.8, 0.8), roughness=0.5, metallic=0.0, specular=0.0, emissive=(0, 0, 0),
             transmission=0.0, ior=1.0, color_tex=-1, roughness_tex=-1, metallic_tex=-1, normal_tex=-1,
             use_normals=False):
    m = np.zeros((), MATERIAL_DTYPE)
    m["color"] = color; m["color_tex"] = color_tex
    m["roughness"] = roughness; m["roughness_tex"] = roughness_tex
    m["metallic"] = metallic; m["metallic_tex"] = metallic_tex
    m["specular"] = specular; m["specular_tex"] = -1
    m["emissive"] = emissive; m["emissive_tex"] = -1
    m["normal"] = (0, 0, 0); m["normal_tex"] = normal_tex
    m["transmission"] = transmission; m["transmission_tex"] = -1
    m["ior"] = ior; m["ior_tex"] = -1
    m["use_normals"] = int(use_normals)
    return m


def make_camera(position, res_x, res_y, subdiv, yfov_deg=39.6, rotation=(0, 0, 0, 1)):
    """`FileLoading.h:678-684`: RectLens(aspect, 1), focal distance 0.5/tan(yfov/2) along +Y."""
    c = np.zeros((), CAMERA_DTYPE)
    c["position"] = position
    c["rotation"] = rotation
    c["focal_position"] = (0.0, np.float32(0.5 / math.tan(0.5 * math.radians(yfov_deg))), 0.0)
    c["lens_width"] = np.float32(res_x / res_y)
    c["lens_height"] = 1.0
    c["res_x"] = res_x; c["res_y"] = res_y; c["subdiv"] = subdiv
    return c


def quat_from_z_rotation(angle):
    """Rotation about +Z (up) as the reference's (x, y, z, w) quaternion."""
    return (0.0, 0.0, math.sin(angle / 2), math.cos(angle / 2))


# ------------------------------------------------------------------ mesh helpers
def _tris_from_arrays(P, UV, mat, N=None):
    """P: (n,3,3) positions, UV: (n,3,2), mat: int or (n,), N: optional (n,3,3) normals."""
    n = P.shape[0]
    t = np.zeros(n, TRI_DTYPE)
    t["p"] = P.reshape(n, 9).astype(np.float32)
    t["uv"] = UV.reshape(n, 6).astype(np.float32)
    t["material"] = mat
    if N is not None:
        t["n"] = N.reshape(n, 9).astype(np.float32)
        t["flags"] = 1
    return t


def grid_mesh(origin, du, dv, nu, nv, mat, rng=None, jitter=0.0, normal_dir=None, uv_scale=1.0, smooth=False):
    """(nu x nv) quad grid spanned by du, dv from origin -> 2*nu*nv triangles.

    Every vertex is displaced along the plane normal by U(-jitter, jitter) (quirk Q3).
    """
    origin = np.asarray(origin, np.float64); du = np.asarray(du, np.float64); dv = np.asarray(dv, np.float64)
    nrm = np.cross(du, dv); nrm /= np.linalg.norm(nrm)
    if normal_dir is not None and np.dot(nrm, normal_dir) < 0:
        nrm = -nrm
    iu, iv = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    V = origin + iu[..., None] * du / nu + iv[..., None] * dv / nv
    if jitter and rng is not None:
        V = V + nrm * rng.uniform(-jitter, jitter, size=iu.shape)[..., None]
    UV = np.stack([iu / nu, iv / nv], -1) * uv_scale
    a = V[:-1, :-1]; b = V[1:, :-1]; c = V[1:, 1:]; d = V[:-1, 1:]
    ua = UV[:-1, :-1]; ub = UV[1:, :-1]; uc = UV[1:, 1:]; ud = UV[:-1, 1:]
    P = np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)])
    U = np.concatenate([np.stack([ua, ub, uc], -2).reshape(-1, 3, 2), np.stack([ua, uc, ud], -2).reshape(-1, 3, 2)])
    N = None
    if smooth:
        N = np.broadcast_to(nrm, P.shape).copy()
    return _tris_from_arrays(P, U, mat, N)


def box_mesh(center, half, mat, rot_z=0.0):
    """Axis box rotated about Z: 12 triangles."""
    c = np.asarray(center, np.float64); h = np.asarray(half, np.float64)
    cs, sn = math.cos(rot_z), math.sin(rot_z)
    R = np.array([[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]])
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64) * h
    corners = corners @ R.T + c
    faces = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    P, U = [], []
    quv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64)
    for f in faces:
        q = corners[list(f)]
        P += [q[[0, 1, 2]], q[[0, 2, 3]]]
        U += [quv[[0, 1, 2]], quv[[0, 2, 3]]]
    return _tris_from_arrays(np.array(P), np.array(U), mat)


def cylinder_mesh(base, radius, height, nseg, nrow, mat, rng=None, jitter=0.0, smooth=True, uv_scale=1.0):
    """Open cylinder along +Z: 2*nseg*nrow triangles with smooth vertex normals."""
    base = np.asarray(base, np.float64)
    ia, iz = np.meshgrid(np.arange(nseg + 1), np.arange(nrow + 1), indexing="ij")
    ang = 2 * np.pi * ia / nseg
    r = radius + (rng.uniform(-jitter, jitter, size=ia.shape) if (rng is not None and jitter) else 0.0)
    if rng is not None and jitter:
        r[-1] = r[0]
    V = np.stack([base[0] + r * np.cos(ang), base[1] + r * np.sin(ang), base[2] + height * iz / nrow], -1)
    Nn = np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], -1)
    UV = np.stack([ia / nseg, iz / nrow], -1) * uv_scale
    def quads(A):
        return A[:-1, :-1], A[1:, :-1], A[1:, 1:], A[:-1, 1:]
    a, b, c, d = quads(V); na, nb, nc, nd = quads(Nn); ua, ub, uc, ud = quads(UV)
    P = np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)])
    N = np.concatenate([np.stack([na, nb, nc], -2).reshape(-1, 3, 3), np.stack([na, nc, nd], -2).reshape(-1, 3, 3)])
    U = np.concatenate([np.stack([ua, ub, uc], -2).reshape(-1, 3, 2), np.stack([ua, uc, ud], -2).reshape(-1, 3, 2)])
    return _tris_from_arrays(P, U, mat, N if smooth else None)


def sphere_mesh(center, radius, nseg, nrow, mat, smooth=True):
    """UV sphere: 2*nseg*nrow triangles (degenerate pole triangles included, as exporters emit them)."""
    center = np.asarray(center, np.float64)
    ia, ib = np.meshgrid(np.arange(nseg + 1), np.arange(nrow + 1), indexing="ij")
    th = 2 * np.pi * ia / nseg
    ph = np.pi * (ib / nrow) * 0.998 + 0.001 * np.pi
    Nn = np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], -1)
    V = center + radius * Nn
    UV = np.stack([ia / nseg, ib / nrow], -1)
    def quads(A):
        return A[:-1, :-1], A[1:, :-1], A[1:, 1:], A[:-1, 1:]
    a, b, c, d = quads(V); na, nb, nc, nd = quads(Nn); ua, ub, uc, ud = quads(UV)
    P = np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)])
    N = np.concatenate([np.stack([na, nb, nc], -2).reshape(-1, 3, 3), np.stack([na, nc, nd], -2).reshape(-1, 3, 3)])
    U = np.concatenate([np.stack([ua, ub, uc], -2).reshape(-1, 3, 2), np.stack([ua, uc, ud], -2).reshape(-1, 3, 2)])
    return _tris_from_arrays(P, U, mat, N if smooth else None)


# ------------------------------------------------------------------ textures
def checker_texture(size, cells, c0, c1):
    y, x = np.mgrid[0:size, 0:size]
    m = (((x * cells) // size + (y * cells) // size) & 1).astype(bool)
    img = np.where(m[..., None], np.array(c1, np.uint8), np.array(c0, np.uint8)).astype(np.uint8)
    return np.ascontiguousarray(img)


def noise_texture(size, rng, base, amp):
    v = rng.integers(-amp, amp + 1, size=(size, size, 1))
    img = np.clip(np.array(base, np.int32) + v, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img)


def brick_texture(size, rng, brick=(150, 70, 55), mortar=(190, 185, 175)):
    y, x = np.mgrid[0:size, 0:size]
    row = (y * 16) // size
    xs = (x + (row & 1) * (size // 16)) % size
    m = (((y * 16) % size) < size // 80 * 16 // 4) | (((xs * 8) % size) < size // 40)
    img = np.where(m[..., None], np.array(mortar, np.int32), np.array(brick, np.int32))
    img = np.clip(img + rng.integers(-12, 13, size=(size, size, 1)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img)


def bump_normal_texture(size, waves=24):
    """Tangent-space normal map stored with the reference's (c-128)/127 transform (Texture.h:116-125,151-153)."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float64)
    dx = 0.35 * np.cos(2 * np.pi * waves * x / size)
    dy = 0.35 * np.cos(2 * np.pi * waves * y / size)
    n = np.stack([-dx, -dy, np.ones_like(dx)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    img = np.clip(n * 127 + 128, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img)


# ------------------------------------------------------------------ scenes
def cornell(res=512, subdiv=4, with_sun=False):
    """C1: Cornell-style box centred on the origin, walls at +-1."""
    mats = np.stack([
        material((0.73, 0.73, 0.73), roughness=0.9),            # 0 white
        material((0.65, 0.05, 0.05), roughness=0.9),            # 1 red
        material((0.12, 0.45, 0.15), roughness=0.9),            # 2 green
        material((0.0, 0.0, 0.0), roughness=0.9, emissive=(25, 25, 25)),  # 3 light
        material((0.8, 0.8, 0.85), roughness=0.15, metallic=1.0),         # 4 metal box
    ])
    def quad(o, du, dv, m):
        return grid_mesh(o, du, dv, 1, 1, m)
    parts = [
        quad((-1, -1, -1), (2, 0, 0), (0, 2, 0), 0),      # floor z=-1
        quad((-1, -1, 1), (2, 0, 0), (0, 2, 0), 0),       # ceiling z=1
        quad((-1, 1, -1), (2, 0, 0), (0, 0, 2), 0),       # back y=1
        quad((-1, -1, -1), (0, 2, 0), (0, 0, 2), 1),      # left x=-1
        quad((1, -1, -1), (0, 2, 0), (0, 0, 2), 2),       # right x=1
        box_mesh((-0.35, 0.3, -0.4), (0.3, 0.3, 0.6), 0, rot_z=0.3),
        box_mesh((0.4, -0.25, -0.7), (0.3, 0.3, 0.3), 4, rot_z=-0.3),
        quad((-0.25, -0.25, 0.995), (0.5, 0, 0), (0, 0.5, 0), 3),   # emissive panel
    ]
    tris = np.concatenate(parts)
    suns = [((0.3, -0.2, 0.93), (2.0, 2.0, 2.0))] if with_sun else []
    cam = make_camera((0.0, -3.5, 0.0), res, res, subdiv)
    return SceneDesc("cornell", tris, mats, [], suns, 10, 32, 0.25, cam)


def atrium(n_tris=262144, res=(1920, 1080), subdiv=8, max_generations=8, seed=7, tex_size=1024,
           monte_carlo_max=32):
    """C2: procedural colonnaded atrium ("Sponza-class"), exactly `n_tris` triangles, 12 materials, 1 sun.

    Long hall along +Y, open roof slot so the sun enters, two colonnades, upper galleries, banners and
    vases; all tessellated surfaces carry a >=1e-4 displacement (quirk Q3).
    """
    rng = np.random.default_rng(seed)
    textures = [
        TextureDesc(brick_texture(tex_size, rng)),                                   # 0
        TextureDesc(checker_texture(tex_size, 16, (200, 195, 180), (90, 85, 80))),   # 1 floor
        TextureDesc(noise_texture(tex_size, rng, (185, 180, 170), 25)),              # 2 stone
        TextureDesc(checker_texture(tex_size, 64, (170, 30, 30), (150, 20, 25))),    # 3 banner red
        TextureDesc(noise_texture(tex_size, rng, (60, 110, 60), 30)),                # 4 green cloth
        TextureDesc(noise_texture(tex_size, rng, (128, 128, 128), 100)),             # 5 roughness map
    ]
    mats = np.stack([
        material(color_tex=0, roughness=0.85),                        # 0 brick walls
        material(color_tex=1, roughness=0.35, roughness_tex=-1),      # 1 polished floor
        material(color_tex=2, roughness=0.7),                         # 2 stone columns
        material(color_tex=3, roughness=0.95),                        # 3 red banners
        material(color_tex=4, roughness=0.95),                        # 4 green banners
        material((0.9, 0.75, 0.3), roughness=0.25, metallic=1.0),     # 5 brass vases
        material((0.75, 0.75, 0.78), roughness=0.1, metallic=1.0),    # 6 steel
        material((0.7, 0.7, 0.7), roughness=0.6, roughness_tex=5),    # 7 plaster w/ roughness map
        material((0.2, 0.3, 0.7), roughness=0.5),                     # 8 blue tiles
        material((0.85, 0.82, 0.78), roughness=0.8),                  # 9 ceiling
        material((0.4, 0.25, 0.15), roughness=0.6),                   # 10 wood
        material((0.95, 0.95, 0.95), roughness=0.3, specular=0.5),    # 11 marble
    ])
    W, L, H = 6.0, 30.0, 10.0     # half-width, half-length, height; floor at z=-1.5 (origin is inside the hall)
    z0 = -1.5
    J = 2e-3
    rng_state = rng.bit_generator.state
    # scale tessellation with the requested size (n_tris=262144 -> s=1); shrink until the base fits
    s = max(0.02, math.sqrt(n_tris / 262144.0))
    while True:
        rng.bit_generator.state = rng_state
        parts = []
        budget = n_tris

        def add(t):
            nonlocal budget
            parts.append(t)
            budget -= len(t)

        g = lambda n: max(1, int(round(n * s)))
        add(grid_mesh((-W, -L, z0), (2 * W, 0, 0), (0, 2 * L, 0), g(64), g(320), 1, rng, J, uv_scale=12))       # floor
        add(grid_mesh((-W, -L, z0), (0, 2 * L, 0), (0, 0, H), g(320), g(48), 0, rng, J, uv_scale=10))           # left wall
        add(grid_mesh((W, -L, z0), (0, 2 * L, 0), (0, 0, H), g(320), g(48), 0, rng, J, uv_scale=10))            # right wall
        add(grid_mesh((-W, L, z0), (2 * W, 0, 0), (0, 0, H), g(64), g(48), 7, rng, J, uv_scale=4))              # far wall
        add(grid_mesh((-W, -L, z0), (2 * W, 0, 0), (0, 0, H), g(64), g(48), 7, rng, J, uv_scale=4))             # near wall
        # roof: two strips leaving a slot open along the hall
        add(grid_mesh((-W, -L, z0 + H), (W - 1.5, 0, 0), (0, 2 * L, 0), g(24), g(160), 9, rng, J))
        add(grid_mesh((1.5, -L, z0 + H), (W - 1.5, 0, 0), (0, 2 * L, 0), g(24), g(160), 9, rng, J))
        # galleries (upper floors along both sides)
        add(grid_mesh((-W, -L, z0 + 4.5), (2.0, 0, 0), (0, 2 * L, 0), g(12), g(160), 10, rng, J, uv_scale=8))
        add(grid_mesh((W - 2.0, -L, z0 + 4.5), (2.0, 0, 0), (0, 2 * L, 0), g(12), g(160), 10, rng, J, uv_scale=8))
        # colonnades
        ncol = 14
        for side in (-1, 1):
            for i in range(ncol):
                y = -L + (i + 0.5) * (2 * L / ncol)
                add(cylinder_mesh((side * (W - 2.2), y, z0), 0.35, 4.5, max(3, g(48)), g(24), 2, rng, 1e-3, uv_scale=2))
                add(cylinder_mesh((side * (W - 2.2), y, z0 + 4.5), 0.25, 5.5, max(3, g(32)), g(16), 11, rng, 1e-3))
        # banners hanging in the slot
        for i in range(10):
            y = -L + (i + 0.5) * (2 * L / 10)
            m = 3 if i % 2 == 0 else 4
            add(grid_mesh((-1.2, y, z0 + 4.0), (2.4, 0, 0.15), (0, 0.05, 4.5), g(24), g(40), m, rng, 4e-3))
        # vases / spheres on the floor
        for i in range(16):
            y = -L + (i + 0.5) * (2 * L / 16)
            x = (-1) ** i * 2.0
            add(sphere_mesh((x, y, z0 + 0.45), 0.45, max(3, g(40)), max(2, g(24)), 5 if i % 3 else 6))
        # blue tiled pool rim boxes
        for i in range(6):
            y = -L + (i + 0.5) * (2 * L / 6)
            add(box_mesh((0.0, y, z0 + 0.15), (0.9, 1.5, 0.15), 8, rot_z=0.05 * i))
        if budget >= 0:
            break
        s *= 0.9
        if s < 1e-3:
            raise ValueError(f"atrium base geometry does not fit n_tris={n_tris}")
    # fill the remaining budget with clutter spheres on the galleries, then trim to the exact count
    k = 0
    while budget > 0:
        side = -1 if k % 2 == 0 else 1
        y = -L + 1.0 + ((k // 2) * 1.7) % (2 * L - 2.0)
        nseg, nrow = max(3, g(32)), max(2, g(20))
        t = sphere_mesh((side * (W - 1.0), y, z0 + 4.5 + 0.3 + 0.002 * k), 0.3, nseg, nrow, 5 + (k % 7))
        if len(t) > budget:
            t = t[:budget]
        add(t)
        k += 1
    tris = np.concatenate(parts)
    assert len(tris) == n_tris
    sun_dir = np.array((0.18, 0.25, 0.95)); sun_dir = sun_dir / np.linalg.norm(sun_dir)
    suns = [(tuple(sun_dir), (3.0, 2.9, 2.7))]
    cam = make_camera((0.4, -L + 2.0, 0.3), res[0], res[1], subdiv, yfov_deg=55.0)
    return SceneDesc("atrium", tris, mats, textures, suns, max_generations, monte_carlo_max, 0.25, cam)


def soup(n_tris=2000, seed=3, extent=5.0, size=0.35, n_materials=4):
    """Random triangle soup in [-extent, extent]^3: adversarial parity scene (every leaf box contains the origin)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_tris, 1, 3))
    P = c + rng.uniform(-size, size, size=(n_tris, 3, 3))
    UV = rng.uniform(0, 1, size=(n_tris, 3, 2))
    mats = np.stack([material(tuple(rng.uniform(0.2, 0.9, 3)), roughness=float(rng.uniform(0.1, 1.0)))
                     for _ in range(n_materials)])
    tris = _tris_from_arrays(P, UV, rng.integers(0, n_materials, n_tris))
    cam = make_camera((0.0, -3.0 * extent, 0.0), 256, 256, 2)
    sun_dir = np.array((0.3, -0.4, 0.86)); sun_dir /= np.linalg.norm(sun_dir)
    return SceneDesc("soup", tris, mats, [], [(tuple(sun_dir), (1.5, 1.5, 1.5))], 4, 8, 0.25, cam)


def interior(n_tris=60000, res=(640, 360), subdiv=4, seed=11, tex_size=512):
    """C3-style room: normal-mapped floor, transmissive spheres (transmission 1, IOR 1.5), textured walls."""
    rng = np.random.default_rng(seed)
    textures = [
        TextureDesc(checker_texture(tex_size, 8, (210, 200, 190), (120, 110, 100))),
        TextureDesc(bump_normal_texture(tex_size), normal_transform=True),
        TextureDesc(brick_texture(tex_size, rng)),
    ]
    mats = np.stack([
        material(color_tex=0, roughness=0.4, normal_tex=1, use_normals=True),   # 0 floor
        material(color_tex=2, roughness=0.9),                                   # 1 walls
        material((0.95, 0.95, 0.95), roughness=0.05, transmission=1.0, ior=1.5),  # 2 glass
        material((0.9, 0.6, 0.2), roughness=0.3, metallic=1.0),                 # 3 copper
        material((0, 0, 0), roughness=0.9, emissive=(12, 11, 10)),              # 4 lamp
        material((0.8, 0.8, 0.8), roughness=0.8),                               # 5 ceiling
    ])
    J = 1e-3
    s = max(0.05, math.sqrt(n_tris / 60000.0))
    g = lambda n: max(1, int(round(n * s)))
    parts = []
    budget = n_tris
    def add(t):
        nonlocal budget
        parts.append(t); budget -= len(t)
    add(grid_mesh((-4, -4, -1.2), (8, 0, 0), (0, 8, 0), g(64), g(64), 0, rng, J, uv_scale=4, smooth=True))
    add(grid_mesh((-4, -4, 2.8), (8, 0, 0), (0, 8, 0), g(32), g(32), 5, rng, J))
    add(grid_mesh((-4, 4, -1.2), (8, 0, 0), (0, 0, 4), g(48), g(24), 1, rng, J, uv_scale=3))
    add(grid_mesh((-4, -4, -1.2), (0, 8, 0), (0, 0, 4), g(48), g(24), 1, rng, J, uv_scale=3))
    add(grid_mesh((4, -4, -1.2), (0, 8, 0), (0, 0, 4), g(48), g(24), 1, rng, J, uv_scale=3))
    add(grid_mesh((-1, -1, 2.75), (2, 0, 0), (0, 2, 0), 2, 2, 4, rng, 1e-4))
    for i in range(5):
        add(sphere_mesh((-2.4 + 1.2 * i, 1.0 + 0.3 * (i % 2), -0.6), 0.55, g(48), g(32), 2 if i % 2 == 0 else 3))
    k = 0
    while budget > 0:
        t = sphere_mesh((-3.2 + 0.8 * (k % 9), -1.5 - 0.6 * (k // 9), -0.95 + 0.001 * k), 0.25, g(24), g(16), 2 + (k % 2))
        if len(t) > budget:
            t = t[:budget]
        add(t); k += 1
    tris = np.concatenate(parts)[:n_tris]
    sun_dir = np.array((0.2, -0.3, 0.93)); sun_dir /= np.linalg.norm(sun_dir)
    cam = make_camera((0.0, -3.6, 0.2), res[0], res[1], subdiv, yfov_deg=60.0)
    return SceneDesc("interior", tris, mats, textures, [(tuple(sun_dir), (1.0, 1.0, 1.0))], 8, 32, 0.25, cam)


def diffuse_box(n_clutter=50000, res=(1920, 1080), subdiv=4, max_generations=16, monte_carlo_max=1, seed=5):
    """C5: closed diffuse box + clutter, roughness 1, emissive panels only (incoherent-ray stress)."""
    rng = np.random.default_rng(seed)
    mats = np.stack([
        material((0.75, 0.75, 0.75), roughness=1.0),
        material((0.75, 0.3, 0.3), roughness=1.0),
        material((0.3, 0.3, 0.75), roughness=1.0),
        material((0, 0, 0), roughness=1.0, emissive=(20, 20, 20)),
    ])
    J = 1e-3
    parts = [
        grid_mesh((-3, -3, -2), (6, 0, 0), (0, 6, 0), 16, 16, 0, rng, J),
        grid_mesh((-3, -3, 2), (6, 0, 0), (0, 6, 0), 16, 16, 0, rng, J),
        grid_mesh((-3, 3, -2), (6, 0, 0), (0, 0, 4), 16, 16, 0, rng, J),
        grid_mesh((-3, -3, -2), (6, 0, 0), (0, 0, 4), 16, 16, 0, rng, J),
        grid_mesh((-3, -3, -2), (0, 6, 0), (0, 0, 4), 16, 16, 1, rng, J),
        grid_mesh((3, -3, -2), (0, 6, 0), (0, 0, 4), 16, 16, 2, rng, J),
        grid_mesh((-1, -1, 1.99), (2, 0, 0), (0, 2, 0), 2, 2, 3, rng, 1e-4),
    ]
    c = rng.uniform(-2.6, 2.6, size=(n_clutter, 1, 3)) * np.array([1, 1, 0.6])
    P = c + rng.uniform(-0.06, 0.06, size=(n_clutter, 3, 3))
    parts.append(_tris_from_arrays(P, rng.uniform(0, 1, size=(n_clutter, 3, 2)), rng.integers(0, 3, n_clutter)))
    tris = np.concatenate(parts)
    cam = make_camera((0.0, -2.9, 0.0), res[0], res[1], subdiv, yfov_deg=70.0)
    return SceneDesc("diffuse_box", tris, mats, [], [], max_generations, monte_carlo_max, 0.25, cam)


def instanced(n_instances=10_000, mesh_tris=1_000, res=(3840, 2160), subdiv=4, max_generations=8, monte_carlo_max=32, seed=13,
              floor=True):
    """C4 as SURVEY §8(d) words it: `n_instances` instances of ONE `mesh_tris`-triangle mesh on a jittered grid,
    FLATTENED into world-space triangles (the reference has no instancing at trace time: `Object::fetchData` applies
    every object's transform to its mesh's triangles, Primitives.h:423-434), n_instances x mesh_tris triangles (+ 8 192
    of a displaced floor unless floor=False). Each instance has its own rotation about Z, scale and jittered position;
    vertex normals are rotated only, as the reference does. The grid is centred on the origin (quirk Q1 applies to every
    leaf), the floor is displaced per vertex (quirk Q3). One sun; 4 materials dealt per instance."""
    rng = np.random.default_rng(seed)
    # the mesh: a dented sphere of exactly mesh_tris triangles (2 x nseg x nrow), smooth normals
    nrow = max(2, int(round(math.sqrt(mesh_tris / 2.0 / 1.25))))
    nseg = max(3, mesh_tris // (2 * nrow))
    base = sphere_mesh((0, 0, 0), 1.0, nseg, nrow, 0)
    if len(base) < mesh_tris:                                            # top up with copies of the first triangles, shrunk inside
        extra = base[: mesh_tris - len(base)].copy()
        extra["p"] *= np.float32(0.5)
        base = np.concatenate([base, extra])
    base = base[:mesh_tris]
    P0 = base["p"].reshape(-1, 3, 3).astype(np.float64)
    r = np.linalg.norm(P0, axis=-1, keepdims=True)
    dent = 1.0 + 0.12 * np.sin(5.0 * P0[..., 0:1]) * np.cos(4.0 * P0[..., 2:3])
    P0 = P0 * dent
    N0 = base["n"].reshape(-1, 3, 3).astype(np.float64)
    side = int(math.ceil(math.sqrt(n_instances)))
    pitch = 1.0
    half = 0.5 * side * pitch
    mats = np.stack([
        material((0.75, 0.75, 0.72), roughness=0.8),
        material((0.8, 0.45, 0.2), roughness=0.4),
        material((0.3, 0.5, 0.8), roughness=0.6),
        material((0.85, 0.85, 0.9), roughness=0.15, metallic=1.0),
        material((0.55, 0.6, 0.5), roughness=0.9),                       # 4 floor
    ])
    tris = np.zeros(n_instances * mesh_tris + (8192 if floor else 0), TRI_DTYPE)
    chunk = 256
    for i0 in range(0, n_instances, chunk):
        i1 = min(n_instances, i0 + chunk)
        k = np.arange(i0, i1)
        gx, gy = k % side, k // side
        jit = rng.uniform(-0.22, 0.22, size=(len(k), 2))
        sc = rng.uniform(0.22, 0.32, size=len(k))
        ang = rng.uniform(0, 2 * np.pi, size=len(k))
        cs, sn = np.cos(ang), np.sin(ang)
        R = np.zeros((len(k), 3, 3)); R[:, 0, 0] = cs; R[:, 0, 1] = -sn; R[:, 1, 0] = sn; R[:, 1, 1] = cs; R[:, 2, 2] = 1
        T = np.stack([(gx + 0.5) * pitch - half + jit[:, 0], (gy + 0.5) * pitch - half + jit[:, 1], sc * 1.05 + 0.01], -1)
        P = np.einsum("iab,tvb->itva", R, P0) * sc[:, None, None, None] + T[:, None, None, :]
        N = np.einsum("iab,tvb->itva", R, N0)
        blk = tris[i0 * mesh_tris:i1 * mesh_tris]
        blk["p"] = P.reshape(-1, 9).astype(np.float32)
        blk["n"] = N.reshape(-1, 9).astype(np.float32)
        blk["uv"] = np.tile(base["uv"], (len(k), 1))
        blk["material"] = np.repeat(k % 4, mesh_tris)
        blk["flags"] = 1
    if floor:
        tris[n_instances * mesh_tris:] = grid_mesh((-half - 2, -half - 2, 0.0), (2 * half + 4, 0, 0), (0, 2 * half + 4, 0), 64, 64, 4, rng, 2e-3)
    sun_dir = np.array((0.35, 0.2, 0.9)); sun_dir /= np.linalg.norm(sun_dir)
    tilt = -0.52
    cam = make_camera((0.3, -half - 1.0, 6.0), res[0], res[1], subdiv, yfov_deg=50.0, rotation=(math.sin(tilt / 2), 0.0, 0.0, math.cos(tilt / 2)))
    return SceneDesc("instanced", tris, mats, [], [(tuple(sun_dir), (2.5, 2.4, 2.2))], max_generations, monte_carlo_max, 0.25, cam)
