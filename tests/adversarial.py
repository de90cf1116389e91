"""Adversarial scenes and ray sets for the traversal kernel's leaf filter (shared by tests/test_leaf_filter_*.py and
tools/exp_leaf_filter.py).

The filter (raytracing_amd/csrc/srx_trace.hip node_test) skips a leaf when the ray misses the padded TIGHT box of its
triangles; the reference (Core.h:176-241) tests every leaf whose origin-grown box the ray enters. The two agree as
long as the reference's fp32 Moller-Trumbore test (Primitives.h:159-202) accepts nothing outside that padded box, which
is what these inputs attack: edge-on rays (det ~ 0, 1/det huge), sliver triangles, large coordinate / feature ratios,
rays along leaf-box faces and rays that start on triangle planes.
"""
import numpy as np

from raytracing_amd import scenes


def _desc(name, P, mats=4):
    n = len(P)
    UV = np.tile(np.array([[0, 0], [1, 0], [0, 1]], np.float32), (n, 1, 1))
    tris = scenes._tris_from_arrays(np.asarray(P, np.float64), UV, np.arange(n) % mats)
    materials = np.array([scenes.material(color=(0.3 + 0.15 * i, 0.5, 0.7 - 0.1 * i)) for i in range(mats)])
    return scenes.SceneDesc(name=name, tris=tris, materials=materials, suns=[((0.3, -0.2, 0.93), (3.0, 3.0, 3.0))],
                            camera=scenes.make_camera((0.0, -6.0, 0.5), 64, 64, 1))


def sliver_soup(n=3000, seed=1, extent=4.0):
    """Triangles with aspect ratios 1e2 .. 1e6 (long edge ~0.5, height down to 5e-7) in general position."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 3))
    a = rng.normal(size=(n, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(a, rng.normal(size=(n, 3))); b /= np.linalg.norm(b, axis=1, keepdims=True)
    length = rng.uniform(0.2, 0.8, (n, 1))
    height = length * 10.0 ** rng.uniform(-6, -2, (n, 1))
    s = rng.uniform(0.0, 1.0, (n, 1))
    P = np.stack([c, c + a * length, c + a * length * s + b * height], 1)
    return _desc("sliver_soup", P)


def offset_soup(n=3000, seed=2, offset=1.0e4, size=1.0e-3):
    """|coordinates| ~ 1e4 with 1e-3 features: every vertex has ~10 mantissa bits left for the feature."""
    rng = np.random.default_rng(seed)
    c = offset + rng.uniform(-0.05, 0.05, (n, 3))
    P = c[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    d = _desc("offset_soup", P)
    d.camera = scenes.make_camera((offset, offset - 1.0, offset), 64, 64, 1)
    return d


def tilted_sheets(seed=3):
    """Tessellated planes in general orientation (near-coplanar leaves: the case where an edge-on ray's u, v are
    rounding noise over a whole leaf), plus a few well-shaped random triangles."""
    rng = np.random.default_rng(seed)
    parts = []
    for k in range(6):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        u = np.cross(n, rng.normal(size=3)); u /= np.linalg.norm(u)
        v = np.cross(n, u)
        org = rng.uniform(-2.0, 2.0, 3) + n * rng.uniform(0.5, 2.0)
        parts.append(scenes.grid_mesh(org - 1.5 * u - 1.5 * v, 3.0 * u, 3.0 * v, 24, 24, k % 4, rng=rng, jitter=2e-4))
    d = scenes.soup(400, seed=seed, extent=3.0)
    d.tris = np.concatenate(parts + [d.tris])
    d.name = "tilted_sheets"
    return d


ADVERSARIAL_SCENES = {
    "sliver_soup": sliver_soup,
    "offset_soup": offset_soup,
    "tilted_sheets": tilted_sheets,
    "atrium_8k": lambda: scenes.atrium(8192, res=(64, 36), subdiv=1, tex_size=32),
}


def adversarial_rays(desc, seed=5, n=60000):
    """{kind: (origins, directions)} float32 for one scene description."""
    with np.errstate(all="ignore"):            # degenerate triangles give 0/0 directions: NaN rays are part of the input
        return _adversarial_rays(desc, seed, n)


def _adversarial_rays(desc, seed, n):
    rng = np.random.default_rng(seed)
    P = desc.tris["p"].reshape(-1, 3, 3).astype(np.float64)
    m = len(P)
    lo, hi = P.reshape(-1, 3).min(0), P.reshape(-1, 3).max(0)
    centre, ext = 0.5 * (lo + hi), float((hi - lo).max())
    out = {}

    # ---- edge-on rays: in the plane of a triangle (tilted out of it by 1e-9 .. 1e-2 rad), passing OUTSIDE it at
    # 1e-6 .. 10 triangle sizes from an edge; the reference's det is rounding noise for the smallest tilts
    k = rng.integers(0, m, n)
    p1, e1, e2 = P[k, 0], P[k, 1] - P[k, 0], P[k, 2] - P[k, 0]
    nrm = np.cross(e1, e2)
    area2 = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm / np.maximum(area2, 1e-300)
    size = np.sqrt(np.maximum(area2, 1e-300))
    edge_dir = e1 / np.linalg.norm(e1, axis=1, keepdims=True)
    outward = np.cross(edge_dir, nrm)                              # in-plane, perpendicular to edge p1p2
    outward *= -np.sign(np.sum(outward * e2, 1, keepdims=True))    # away from the third vertex
    along = rng.uniform(-0.5, 1.5, (n, 1))
    gap = size * 10.0 ** rng.uniform(-6, 1, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1), p=[0.2, 0.8])
    X = p1 + e1 * along + outward * gap                             # where the ray crosses the plane
    phi = rng.uniform(0, 2 * np.pi, (n, 1))
    inplane = edge_dir * np.cos(phi) + outward * np.sin(phi)
    tilt = 10.0 ** rng.uniform(-9, -2, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    d = inplane + nrm * tilt
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = ext * 10.0 ** rng.uniform(-2, 0.5, (n, 1))
    out["edge_on"] = ((X - d * dist).astype(np.float32), d.astype(np.float32))

    # ---- the same construction aimed at the filter's own margins: tilts around the guard's threshold (sin(gamma)|cos(phi)| = 1/80,
    # i.e. 5e-3 .. 0.1 rad out of the plane) and gaps of 1e-5 .. 3e-3 of the distance flown, around the relative pad (5e-4)
    k = rng.integers(0, m, n // 2)
    p1, e1, e2 = P[k, 0], P[k, 1] - P[k, 0], P[k, 2] - P[k, 0]
    nrm = np.cross(e1, e2)
    nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    edge_dir = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-300)
    outward = np.cross(edge_dir, nrm)
    outward *= -np.sign(np.sum(outward * e2, 1, keepdims=True))
    dist = ext * 10.0 ** rng.uniform(-1.5, 0.7, (n // 2, 1))
    gap = dist * 10.0 ** rng.uniform(-5, -2.5, (n // 2, 1))
    X = p1 + e1 * rng.uniform(0.0, 1.0, (n // 2, 1)) + outward * gap
    phi = rng.uniform(0, 2 * np.pi, (n // 2, 1))
    tilt = 10.0 ** rng.uniform(np.log10(5e-3), -1, (n // 2, 1)) * rng.choice([-1.0, 1.0], (n // 2, 1))
    d = edge_dir * np.cos(phi) + outward * np.sin(phi) + nrm * tilt
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out["near_threshold"] = ((X - d * dist).astype(np.float32), d.astype(np.float32))

    # ---- rays that START on a triangle's plane (tvec in the plane) and leave at a shallow angle
    k = rng.integers(0, m, n // 2)
    p1, e1, e2 = P[k, 0], P[k, 1] - P[k, 0], P[k, 2] - P[k, 0]
    nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    uv = rng.uniform(-1.0, 2.0, (n // 2, 2))
    o = p1 + e1 * uv[:, :1] + e2 * uv[:, 1:]
    d = e1 * rng.normal(size=(n // 2, 1)) + e2 * rng.normal(size=(n // 2, 1))
    d = d / np.linalg.norm(d, axis=1, keepdims=True) + nrm * (10.0 ** rng.uniform(-8, -1, (n // 2, 1)))
    out["on_plane"] = (o.astype(np.float32), d.astype(np.float32))

    # ---- rays along the faces of triangle bounding boxes (what leaf boxes are made of): axis-parallel and nearly so,
    # at a vertex coordinate +- a few ulps
    k = rng.integers(0, m, n // 2)
    vtx = P[k, rng.integers(0, 3, n // 2)]
    ax = rng.integers(0, 3, n // 2)
    d = rng.normal(size=(n // 2, 3)) * (10.0 ** rng.uniform(-7, -2, (n // 2, 1)))
    d[np.arange(n // 2), ax] = rng.choice([-1.0, 1.0], n // 2)
    d[:: 5] = np.where(np.abs(d[::5]) < 0.5, 0.0, d[::5])           # every fifth exactly axis-parallel
    o32 = (vtx - d * ext * rng.uniform(0.1, 1.5, (n // 2, 1))).astype(np.float32)
    other = (ax + 1 + rng.integers(0, 2, n // 2)) % 3
    bump = rng.integers(-3, 4, n // 2)
    o32[np.arange(n // 2), other] = np.nextafter(vtx[np.arange(n // 2), other].astype(np.float32),
                                                 np.where(bump >= 0, np.float32(np.inf), np.float32(-np.inf)))
    out["box_faces"] = (o32, d.astype(np.float32))

    # ---- far origins (|o| up to 1e4 scene sizes) aimed at the geometry: large tvec, large rounding in o - p1
    k = rng.integers(0, m, n // 2)
    target = P[k].mean(1) + rng.normal(size=(n // 2, 3)) * 0.3 * np.sqrt(np.linalg.norm(np.cross(P[k, 1] - P[k, 0], P[k, 2] - P[k, 0]), axis=1, keepdims=True))
    dirn = rng.normal(size=(n // 2, 3)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    far = ext * 10.0 ** rng.uniform(0, 4, (n // 2, 1))
    out["far_origin"] = ((target - dirn * far).astype(np.float32), dirn.astype(np.float32))

    # ---- ordinary incoherent rays through the scene, for scale
    o = (centre + rng.uniform(-0.6, 0.6, (n // 2, 3)) * ext)
    dd = rng.normal(size=(n // 2, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    out["incoherent"] = (o.astype(np.float32), dd.astype(np.float32))
    return out
