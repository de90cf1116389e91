"""oracle/sro_gltf.py — CPU restatement of the reference's scene ingestion for small files.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). PARITY UNPINNED: the reference reads .glb files through tinygltf
and stb_image (third-party headers in its tree, not built here: FileLoading.h needs SceneManager.h -> SFML, OpenColorIO)
and ships no scene asset, so nothing of this stage can be checked against the reference's own output.

Follows FileManager::parseGLTFData (FileLoading.h:438-782), Tri (Primitives.h:237-252), Object::fetchData /
Packers::transformT / compile_scale / compile_transforms (Primitives.h:347-373, 423-484), Transformation::stack /
collpase (Transformation.h:28-57), Quat::applyRotation / multiply (XYZ.cpp:626-633, 679-683), the XYZ / Quat swizzle
constructors (XYZ.cpp:97, 115-121, 613), Texture::load / commit_char / export_channel (Texture.h:36-67, 116-136,
214-226) and ImageHandler::apply_gamma (ImageProcessing.h:128-133). float32 arithmetic is done with numpy scalars, one
rounding per written operation. Same deliberate differences as the product (csrc/srx_gltf.cpp header)."""
from __future__ import annotations

import json
import math
import struct
import zlib

import numpy as np

F = np.float32


def parse_glb(path):
    b = open(path, "rb").read()
    magic, version, _ = struct.unpack_from("<III", b, 0)
    if magic != 0x46546C67 or version != 2:
        raise ValueError("not a .glb version 2 file")
    at, js, binc = 12, None, b""
    while at + 8 <= len(b):
        n, ty = struct.unpack_from("<II", b, at)
        body = b[at + 8:at + 8 + n]
        if ty == 0x4E4F534A:
            js = json.loads(body.decode("utf-8"))
        elif ty == 0x004E4942 and not binc:
            binc = body
        at += 8 + n
    return js, binc


# ---- XYZ.cpp arithmetic, float32 per operation
def v3(x, y, z):
    return np.array([x, y, z], F)


def cross(a, b):
    return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def normalize(v):
    ln = np.sqrt(F(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
    with np.errstate(all="ignore"):
        return v3(v[0] / ln, v[1] / ln, v[2] / ln)


def apply_rotation(p, r):
    rv = r[:3]
    t = cross(rv, p) * F(2)
    return (p + t * r[3]) + cross(rv, t)


def q_multiply(q2, q1):
    return np.array([
        q1[3] * q2[0] + q1[0] * q2[3] + q1[1] * q2[2] - q1[2] * q2[1],
        q1[3] * q2[1] - q1[0] * q2[2] + q1[1] * q2[3] + q1[2] * q2[0],
        q1[3] * q2[2] + q1[0] * q2[1] - q1[1] * q2[0] + q1[2] * q2[3],
        q1[3] * q2[3] - q1[0] * q2[0] - q1[1] * q2[1] - q1[2] * q2[2]], F)


def collapse_transforms(ts):
    """Transformation::collpase (Transformation.h:47-56) over a chain of ("move", xyz) / ("rot", quat) steps, each
    stacked with Transformation::stack (:24-33): a move adds to the position, a rotation multiplies the quaternion and
    rotates the position so far. Returns (position, rotation)."""
    pos, rot = v3(0, 0, 0), np.array([0, 0, 0, 1], F)
    for kind, val in ts:
        if kind == "move":
            pos = pos + val
        else:
            rot = q_multiply(rot, val)
            pos = apply_rotation(pos, val)
    return pos, rot


def apply_transform(p, pos, rot):
    """Transformation::apply (Transformation.h:35-42): rotate, then translate."""
    return apply_rotation(p, rot) + pos


def num(a, i):
    return float(a[i]) if i < len(a) and isinstance(a[i], (int, float)) else 0.0


def swz_pos(a):
    return v3(F(num(a, 0)), F(-1 * num(a, 2)), F(1 * num(a, 1)))


def swz_scale(a):
    return v3(F(num(a, 0)), F(num(a, 2)), F(num(a, 1)))


def swz_rot(a):
    return np.array([F(num(a, 0)), F(-1 * num(a, 2)), F(1 * num(a, 1)), F(1 * num(a, 3))], F)


def swz(v):
    return v3(F(1) * v[0], F(-1) * v[2], F(1) * v[1])


# ---- images and textures
# JPEG pixels are not restated here: tests hand in what the reference's own stb_image gives for the file (committed under
# tests/golden/jpeg_cases.npz, or oracle/_ref/libref_stb.so) through this hook: bytes -> (h, w, 4) uint8
jpeg_decoder = None


def decode_image(data):
    if data[:2] == b"\xff\xd8":
        if jpeg_decoder is None:
            raise ValueError("JPEG image and no sro_gltf.jpeg_decoder set")
        return jpeg_decoder(bytes(data))
    return decode_png(data)


def decode_png(data):
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    at, idat, plte = 8, b"", b""
    while at + 12 <= len(data):
        n, ty = struct.unpack_from(">I4s", data, at)
        body = data[at + 8:at + 8 + n]
        if ty == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and interlace == 0
        elif ty == b"PLTE":
            plte = body
        elif ty == b"IDAT":
            idat += body
        elif ty == b"IEND":
            break
        at += 12 + n
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    img = np.zeros((h, w * ch), np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = img[y - 1] if y else np.zeros(w * ch, np.int32)
        cur = img[y]
        for x in range(w * ch):
            a = cur[x - ch] if x >= ch else 0
            b = up[x]
            c = up[x - ch] if x >= ch else 0
            if f == 0:
                p = 0
            elif f == 1:
                p = a
            elif f == 2:
                p = b
            elif f == 3:
                p = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[x] = (line[x] + p) & 255
    img = img.astype(np.uint8).reshape(h, w, ch)
    out = np.full((h, w, 4), 255, np.uint8)
    if ctype == 0:
        out[..., :3] = img[..., :1]
    elif ctype == 4:
        out[..., :3] = img[..., :1]; out[..., 3] = img[..., 1]
    elif ctype == 3:
        pal = np.frombuffer(plte, np.uint8).reshape(-1, 3)
        out[..., :3] = pal[img[..., 0]]
    else:
        out[..., :ch] = img
    return out                                           # RGBA, what tinygltf hands over (component = 4)


def gamma_srgb(v):                                       # apply_gamma(in, 1 / 2.2)
    gamma = F(1 / 2.2)
    lums = F(F(v[0] * F(0.2126) + v[1] * F(0.7152)) + v[2] * F(0.0722))
    if lums == 0:
        return v3(0, 0, 0)
    post = F(math.pow(float(lums), 1.0 / float(gamma)))
    return v3(v[0] * post / lums, v[1] * post / lums, v[2] * post / lums)


def uchar(v):
    return np.uint8(int(v))                              # truncation


def load_texture(rgba, srgb, with_scale, scale_v):
    h, w, comp = rgba.shape
    flat = np.concatenate([rgba.reshape(-1), np.zeros(8, np.uint8)])
    out = np.zeros((h, w, 3), np.uint8)
    position = 0
    for y in range(h):
        for x in range(w):
            position += comp                             # Texture.h:41,57: before the read
            v = v3(F(flat[position]) / F(255.0), F(flat[position + 1]) / F(255.0), F(flat[position + 2]) / F(255.0))
            if srgb:
                v = gamma_srgb(v)
            if with_scale:
                nv = normalize((v * F(2) - F(1)) * v3(scale_v, scale_v, 1.0))
                out[y, x] = [uchar(nv[0] * F(127) + F(128)), uchar(nv[1] * F(127) + F(128)), uchar(nv[2] * F(127) + F(128))]
            else:
                out[y, x] = [uchar(v[0] * F(255)), uchar(v[1] * F(255)), uchar(v[2] * F(255))]
    return out


def export_channel(tex, channel):
    out = np.zeros_like(tex)
    h, w, _ = tex.shape
    for y in range(h):
        for x in range(w):
            out[y, x, 0] = uchar(F(F(tex[y, x, channel]) / F(255.0)) * F(255))
    return out


def tex_transform(info):
    tt = (info or {}).get("extensions", {}).get("KHR_texture_transform")
    if tt is None:
        return F(1), F(1)
    sc = tt.get("scale", [])
    return F(num(sc, 0)), F(num(sc, 1))


def load(path):
    """Returns dict(tris=[(p9, uv6, n9, material)], materials=[dict], textures=[dict(rgb, sx, sy, normal_transform)],
    suns=[(dir3, emission3)], camera=dict)."""
    js, binc = parse_glb(path)

    def view(acc_i):
        acc = js["accessors"][acc_i]
        bv = js["bufferViews"][acc["bufferView"]]
        nc = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[acc["type"]]
        dt = {5126: "<f4", 5125: "<u4", 5123: "<u2"}[acc["componentType"]]
        off = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
        stride = bv.get("byteStride", np.dtype(dt).itemsize * nc)
        return np.array([np.frombuffer(binc, dt, nc, off + i * stride) for i in range(acc["count"])]).reshape(acc["count"], nc)

    base_cache = {}

    def base_texture(t_index, srgb, with_scale, scale_v):
        key = (t_index, srgb, with_scale, float(scale_v) if with_scale else None)
        if key not in base_cache:
            im = js["images"][js["textures"][t_index]["source"]]
            bv = js["bufferViews"][im["bufferView"]]
            data = binc[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
            base_cache[key] = load_texture(decode_image(data), srgb, with_scale, scale_v)
        return base_cache[key]

    textures, materials, uv_map = [], [], []

    def add_tex(rgb, sx, sy, nt=False):
        textures.append(dict(rgb=rgb, sx=sx, sy=sy, normal_transform=nt))
        return len(textures) - 1

    for md in js.get("materials", []):
        pbr = md.get("pbrMetallicRoughness", {})
        bc = pbr.get("baseColorFactor")
        mf, rf = pbr.get("metallicFactor", 1.0), pbr.get("roughnessFactor", 1.0)
        em = md.get("emissiveFactor")
        m = dict(color=[F(num(bc, k)) if bc is not None else F(1) for k in range(3)], roughness=F(rf), metallic=F(mf),
                 emissive=[(F(num(em, k)) if em is not None else F(0)) * F(25) for k in range(3)],
                 color_tex=-1, roughness_tex=-1, metallic_tex=-1, normal_tex=-1, use_normals=0)
        uv_index = 0
        bct = pbr.get("baseColorTexture")
        if bct is not None and bct.get("index", -1) != -1:
            uv_index = bct.get("texCoord", 0)
            m["color_tex"] = add_tex(base_texture(bct["index"], True, False, 0), *tex_transform(bct))
        mrt = pbr.get("metallicRoughnessTexture")
        if mrt is not None and mrt.get("index", -1) != -1:
            uv_index = bct.get("texCoord", 0) if bct is not None else 0
            t = base_texture(mrt["index"], False, False, 0)
            sx, sy = tex_transform(mrt)
            metallic, roughness = export_channel(t, 2), export_channel(t, 1)
            if mf != 0:
                if mf != 1:
                    flat = metallic.reshape(-1)
                    for i in range(t.shape[0] * t.shape[1]):
                        flat[i] = np.uint8(int(int(flat[i]) * mf))
                m["metallic_tex"] = add_tex(metallic, sx, sy)
            if rf != 0:
                m["roughness_tex"] = add_tex(roughness, sx, sy)
        nt = md.get("normalTexture")
        if nt is not None and nt.get("index", -1) != -1:
            uv_index = nt.get("texCoord", 0)
            s = F(nt.get("scale", 1.0))
            m["normal_tex"] = add_tex(base_texture(nt["index"], False, True, s), *tex_transform(nt), nt=True)
            m["use_normals"] = 1
        ext = md.get("extensions", {})
        if "KHR_materials_specular" in ext:
            f = ext["KHR_materials_specular"].get("specularColorFactor")
            m["specular"] = F(num(f, 0)) if f is not None else F(0)
        else:
            m["specular"] = F(1)
        m["ior"] = F(ext["KHR_materials_ior"].get("ior", 0.0)) if "KHR_materials_ior" in ext else F(1.5)
        m["transmission"] = F(ext["KHR_materials_transmission"].get("transmissionFactor", 0.0)) if "KHR_materials_transmission" in ext else F(0)
        materials.append(m)
        uv_map.append(uv_index)
    materials.append(dict(color=[F(1)] * 3, roughness=F(0.5), metallic=F(0), emissive=[F(0)] * 3, specular=F(0), ior=F(1.5),
                          transmission=F(0), color_tex=-1, roughness_tex=-1, metallic_tex=-1, normal_tex=-1, use_normals=0))
    uv_map.append(0)
    default_index = len(materials) - 1

    meshes = []
    for md in js.get("meshes", []):
        tris = []
        for pr in md["primitives"]:
            attr = pr["attributes"]
            mat = pr.get("material", -1)
            if mat <= -1:
                mat = default_index
            P = view(attr["POSITION"]).astype(F)
            ta, na = attr.get("TEXCOORD_%d" % uv_map[mat], 0), attr.get("NORMAL", 0)
            UV = view(ta).astype(F) if ta != 0 else None
            N = view(na).astype(F) if na != 0 else None
            idx = view(pr["indices"]).reshape(-1).astype(np.int64)
            for i in range(len(idx) // 3):
                l = idx[3 * i:3 * i + 3]
                uv = np.zeros(6, F)
                if UV is not None:
                    for k in range(3):
                        uv[2 * k] = F(0) - (F(-1) * UV[l[k], 0])
                        uv[2 * k + 1] = F(1) - (F(1) * UV[l[k], 1])
                p1, p2, p3 = swz(P[l[0]]), swz(P[l[1]]), swz(P[l[2]])
                normal = normalize(cross(p3 - p1, p2 - p1))
                n = [normal, normal, normal] if N is None else [swz(N[l[0]]), swz(N[l[1]]), swz(N[l[2]])]
                tris.append((p1, p2, p3, uv, n, mat))
        meshes.append(tris)

    cameras = []
    for cd in js.get("cameras", []):
        p = cd.get("perspective", {})
        cameras.append(dict(position=v3(0, 0, 0), rotation=np.array([0, 0, 0, 1], F), aspect=F(p.get("aspectRatio", 0.0)),
                            distance=F(0.5 / math.tan(0.5 * p.get("yfov", 0.0)))))
    light_defs = js.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    objects, lights, lookup = [], [], {}
    nodes = js.get("nodes", [])
    for ni, nd in enumerate(nodes):
        sc, tr, ro = nd.get("scale"), nd.get("translation"), nd.get("rotation")
        cam, mesh = nd.get("camera", -1), nd.get("mesh", -1)
        light = nd.get("extensions", {}).get("KHR_lights_punctual", {}).get("light", -1)
        if cam > -1:
            if tr:
                cameras[cam]["position"] = cameras[cam]["position"] + swz_pos(tr)
            cameras[cam]["rotation"] = swz_rot(ro) if ro else np.zeros(4, F)
            lookup[ni] = (0, cam)
        elif mesh > -1:
            objects.append(dict(parent=-1, origin=v3(0, 0, 0), scale=swz_scale(sc) if sc else v3(1, 1, 1),
                                transforms=[("rot", swz_rot(ro) if ro else np.array([0, 0, 0, 1], F)),
                                            ("move", swz_pos(tr) if tr else v3(0, 0, 0))], children=[], mesh=mesh))
            lookup[ni] = (1, len(objects) - 1)
        elif light > -1:
            ld = light_defs[light]
            rot = swz_rot(ro) if ro else np.array([0, 0, 0, 1], F)
            assert ld["type"] == "directional"
            d = apply_rotation(v3(0, -1, 0), rot)
            col = ld.get("color")
            inten = F(ld.get("intensity", 1.0))
            e = [(F(num(col, k)) if col is not None else F(1)) * inten / F(50000) for k in range(3)]
            lights.append((d, np.array(e, F)))
            lookup[ni] = (2, len(lights) - 1)
    for ni, (ty, oi) in lookup.items():
        if ty != 1:
            continue
        for c in nodes[ni].get("children", []):
            if c in lookup and lookup[c][0] == 1:
                objects[oi]["children"].append(lookup[c][1])
                objects[lookup[c][1]]["parent"] = oi

    def compile_scale(oi, s):
        s = s * objects[oi]["scale"]
        if objects[oi]["parent"] >= 0:
            s = s * compile_scale(objects[oi]["parent"], s)
        return s

    def compile_transforms(oi, out):
        out.extend(objects[oi]["transforms"])
        if objects[oi]["parent"] >= 0:
            compile_transforms(objects[oi]["parent"], out)

    out_tris = []

    def fetch(oi):
        o = objects[oi]
        s = compile_scale(oi, v3(1, 1, 1))
        ts = []
        compile_transforms(oi, ts)
        pos, rot = collapse_transforms(ts)
        for p1, p2, p3, uv, n, mat in meshes[o["mesh"]]:
            ps = []
            for p in (p1, p2, p3):
                q = (p - o["origin"]) * s
                q = apply_transform(q, pos, rot)
                ps.append(q + o["origin"])
            ns = [apply_rotation(x, rot) for x in n]
            out_tris.append((np.concatenate(ps), uv, np.concatenate(ns), mat))
        for c in o["children"]:
            fetch(c)

    camera, suns, roots = None, [], []
    last = max(lookup) if lookup else -1
    for i in range(last + 1):
        ty, idx = lookup.get(i, (0, 0))
        if ty == 0 and camera is None and cameras:
            camera = idx
        if ty == 1 and objects[idx]["parent"] < 0:
            roots.append(idx)
        if ty == 2:
            suns.append(lights[idx])
    for r in roots:
        fetch(r)
    c = cameras[camera]
    cam = dict(position=c["position"], rotation=c["rotation"], focal_position=v3(0, c["distance"], 0), lens_width=c["aspect"],
               lens_height=F(1))
    return dict(tris=out_tris, materials=materials, textures=textures, suns=suns, camera=cam)
