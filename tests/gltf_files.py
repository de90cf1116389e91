"""Writes small binary glTF files for the ingestion tests (there is no scene asset offline): meshes with and without
normals / uvs, u16 and u32 indices, an interleaved (strided) vertex buffer, nodes with TRS and a child, a plain group
node, a camera, a KHR_lights_punctual sun, materials with the KHR extensions the reference reads, PNG textures."""
import json
import struct
import zlib

import numpy as np


def png_bytes(img, filters=(0, 1, 2, 3, 4)):
    """8-bit PNG of an (h, w, c) array, c = 1 (gray), 3 (RGB) or 4 (RGBA); rows cycle through the given filter types."""
    h, w, c = img.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    rows = img.reshape(h, w * c).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        f = filters[y % len(filters)]
        cur, up = rows[y], (rows[y - 1] if y else np.zeros(w * c, np.int32))
        out = np.zeros(w * c, np.int32)
        for x in range(w * c):
            a = cur[x - c] if x >= c else 0
            b = up[x]
            cc = up[x - c] if x >= c else 0
            if f == 0:
                p = 0
            elif f == 1:
                p = a
            elif f == 2:
                p = b
            elif f == 3:
                p = (a + b) >> 1
            else:
                pp = a + b - cc
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - cc)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
            out[x] = (cur[x] - p) & 255
        raw.append(f)
        raw += bytes(out.astype(np.uint8))

    def chunk(ty, body):
        return struct.pack(">I", len(body)) + ty + body + struct.pack(">I", zlib.crc32(ty + body))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def png_bytes_general(samples, depth, ctype, interlace=False, plte=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """Any PNG: `samples` (h, w, channels) integers at `depth` bits (1, 2, 4, 8, 16), colour type 0 / 2 / 3 / 4 / 6,
    optional Adam7 interlacing, palette and tRNS chunk bodies; every row takes the next of the given filter types."""
    h, w, c = samples.shape
    assert c == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    fb = 2 * c if depth == 16 else (c if depth == 8 else 1)

    def pack(row):                                      # (w', c) samples -> bytes
        flat = row.reshape(-1).astype(np.int64)
        if depth == 16:
            return np.stack([flat >> 8, flat & 255], -1).reshape(-1).astype(np.int32)
        if depth == 8:
            return flat.astype(np.int32)
        per = 8 // depth
        pad = (-len(flat)) % per
        flat = np.concatenate([flat, np.zeros(pad, np.int64)]).reshape(-1, per)
        shifts = np.array([8 - depth * (k + 1) for k in range(per)])
        return (flat << shifts).sum(1).astype(np.int32)

    raw = bytearray()
    row_no = 0
    passes = [(0, 0, 1, 1)] if not interlace else list(zip((0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)))
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for y in range(sub.shape[0]):
            cur = pack(sub[y])
            f = filters[row_no % len(filters)]; row_no += 1
            up = prev if prev is not None else np.zeros_like(cur)
            out = np.zeros_like(cur)
            for x in range(len(cur)):
                a = cur[x - fb] if x >= fb else 0
                b = up[x]
                cc = up[x - fb] if x >= fb else 0
                if f == 0:
                    p = 0
                elif f == 1:
                    p = a
                elif f == 2:
                    p = b
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pp = a + b - cc
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - cc)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                out[x] = (cur[x] - p) & 255
            raw.append(f)
            raw += bytes(out.astype(np.uint8))
            prev = cur

    def chunk(ty, body):
        return struct.pack(">I", len(body)) + ty + body + struct.pack(">I", zlib.crc32(ty + body))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    data = zlib.compress(bytes(raw), 6)
    half = len(data) // 2                                # two IDAT chunks: the stream continues across them
    return out + chunk(b"IDAT", data[:half]) + chunk(b"IDAT", data[half:]) + chunk(b"IEND", b"")


class GlbWriter:
    def __init__(self):
        self.bin = bytearray()
        self.js = {"asset": {"version": "2.0"}, "buffers": [{}], "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                   "materials": [], "cameras": [], "scenes": [{"nodes": []}], "scene": 0, "textures": [], "images": []}

    def view(self, data, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.bin += data
        self.js["bufferViews"].append(v)
        return len(self.js["bufferViews"]) - 1

    def accessor(self, arr, kind, view=None, offset=0):
        arr = np.ascontiguousarray(arr)
        ct = {np.dtype("float32"): 5126, np.dtype("uint32"): 5125, np.dtype("uint16"): 5123}[arr.dtype]
        if view is None:
            view = self.view(arr.tobytes())
        a = {"bufferView": view, "componentType": ct, "count": len(arr), "type": kind}
        if offset:
            a["byteOffset"] = offset
        self.js["accessors"].append(a)
        return len(self.js["accessors"]) - 1

    def image(self, img, **kw):
        self.js["images"].append({"bufferView": self.view(png_bytes(img, **kw)), "mimeType": "image/png", "name": "img%d" % len(self.js["images"])})
        self.js["textures"].append({"source": len(self.js["images"]) - 1})
        return len(self.js["textures"]) - 1

    def write(self, path):
        self.js["buffers"][0]["byteLength"] = len(self.bin)
        js = json.dumps(self.js).encode()
        js += b" " * (-len(js) % 4)
        binc = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(binc)
        with open(path, "wb") as fh:
            fh.write(struct.pack("<III", 0x46546C67, 2, total))
            fh.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            fh.write(struct.pack("<II", len(binc), 0x004E4942) + binc)


def cube(rng, n_quads=6):
    """A unit cube: 24 vertices (positions, normals, uvs), 12 triangles."""
    P, N, UV, I = [], [], [], []
    axes = [(0, 1, 2), (1, 2, 0), (2, 0, 1)]
    for a, (i, j, k) in enumerate(axes):
        for sgn in (-1.0, 1.0):
            base = len(P)
            for (u, v) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                p = [0.0, 0.0, 0.0]; p[i] = sgn * 0.5; p[j] = u * 0.5; p[k] = v * 0.5
                n = [0.0, 0.0, 0.0]; n[i] = sgn
                P.append(p); N.append(n); UV.append([(u + 1) / 2, (v + 1) / 2])
            I += [base, base + 1, base + 2, base, base + 2, base + 3] if sgn > 0 else [base, base + 2, base + 1, base, base + 3, base + 2]
    P = np.array(P, np.float32) + rng.uniform(-1e-3, 1e-3, (24, 3)).astype(np.float32)
    return P, np.array(N, np.float32), np.array(UV, np.float32), np.array(I)


def write_test_scene(path, seed=3, textures=True):
    rng = np.random.default_rng(seed)
    g = GlbWriter()
    g.accessor(np.zeros((1, 3), np.float32), "VEC3")          # accessor 0 is never an attribute (index 0 reads as "absent")
    # materials
    if textures:
        t_col = g.image(rng.integers(0, 256, (6, 5, 3), dtype=np.uint8).astype(np.uint8))
        t_mr = g.image(rng.integers(0, 256, (4, 4, 4), dtype=np.uint8).astype(np.uint8), filters=(4, 3, 1))
        t_nrm = g.image(rng.integers(0, 256, (3, 7, 3), dtype=np.uint8).astype(np.uint8), filters=(2,))
        g.js["materials"].append({
            "name": "textured", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "metallicFactor": 0.5, "roughnessFactor": 0.75,
                                                          "baseColorTexture": {"index": t_col, "extensions": {"KHR_texture_transform": {"scale": [2.0, 3.0], "offset": [0.1, 0.2]}}},
                                                          "metallicRoughnessTexture": {"index": t_mr}},
            "normalTexture": {"index": t_nrm, "scale": 0.8}, "emissiveFactor": [0.1, 0.0, 0.3]})
    else:
        g.js["materials"].append({"name": "plain", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "metallicFactor": 0.5, "roughnessFactor": 0.75},
                                  "emissiveFactor": [0.1, 0.0, 0.3]})
    g.js["materials"].append({"name": "glass", "pbrMetallicRoughness": {"roughnessFactor": 0.05, "metallicFactor": 0.0},
                              "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.45},
                                             "KHR_materials_specular": {"specularColorFactor": [0.5, 0.6, 0.7]}}})
    g.js["materials"].append({"name": "spec_no_factor", "extensions": {"KHR_materials_specular": {"specularFactor": 0.3}}})
    # mesh 0: cube with normals + uvs, u16 indices, material 0
    P, N, UV, I = cube(rng)
    g.js["meshes"].append({"name": "cube", "primitives": [{"attributes": {"POSITION": g.accessor(P, "VEC3"), "NORMAL": g.accessor(N, "VEC3"),
                                                                              "TEXCOORD_0": g.accessor(UV, "VEC2")},
                                                           "indices": g.accessor(I.astype(np.uint16), "SCALAR"), "material": 0}]})
    # mesh 1: two primitives; an interleaved position+normal buffer (stride 24), u32 indices, glass; then a flat quad, no material
    inter = np.concatenate([P, N], axis=1).astype(np.float32)
    v = g.view(inter.tobytes(), stride=24)
    quad = np.array([[-2, 0.01, -2], [2, 0.02, -2], [2, -0.01, 2], [-2, 0.0, 2]], np.float32)
    g.js["meshes"].append({"name": "two", "primitives": [
        {"attributes": {"POSITION": g.accessor(inter[:, :3], "VEC3", view=v), "NORMAL": g.accessor(inter[:, 3:], "VEC3", view=v, offset=12)},
         "indices": g.accessor(I.astype(np.uint32), "SCALAR"), "material": 1},
        {"attributes": {"POSITION": g.accessor(quad, "VEC3")}, "indices": g.accessor(np.array([0, 1, 2, 0, 2, 3], np.uint16), "SCALAR")}]})
    # nodes: group (ignored transform) -> cube node (TRS) -> child cube (scaled); camera; sun; second mesh; a material-2 cube
    q = np.array([0.1, 0.2, 0.3, 0.9]); q = (q / np.linalg.norm(q)).tolist()
    g.js["nodes"] = [
        {"name": "group", "translation": [10, 10, 10], "children": [1]},
        {"name": "cube", "mesh": 0, "translation": [0.5, 1.0, -2.0], "rotation": q, "scale": [1.0, 2.0, 0.5], "children": [2]},
        {"name": "child", "mesh": 0, "translation": [0.0, 1.5, 0.0], "scale": [0.5, 0.25, 0.75]},
        {"name": "cam", "camera": 0, "translation": [0.0, 1.0, 6.0], "rotation": [-0.0871557, 0.0, 0.0, 0.9961947]},
        {"name": "sun", "rotation": [0.3826834, 0.0, 0.0, 0.9238795], "extensions": {"KHR_lights_punctual": {"light": 0}}},
        {"name": "two", "mesh": 1, "translation": [-1.0, 0.0, 0.0]},
        {"name": "no_trs", "mesh": 0},
    ]
    g.js["cameras"].append({"type": "perspective", "perspective": {"aspectRatio": 1.5, "yfov": 0.6911, "znear": 0.1}})
    g.js["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "directional", "color": [1.0, 0.9, 0.8], "intensity": 3.5}]}}
    g.js["extensionsUsed"] = ["KHR_lights_punctual", "KHR_materials_transmission", "KHR_materials_ior", "KHR_materials_specular", "KHR_texture_transform"]
    g.js["scenes"][0]["nodes"] = [0, 3, 4, 5, 6]
    g.write(path)
    return g
