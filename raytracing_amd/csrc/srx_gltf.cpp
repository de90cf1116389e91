// raytracing_amd/csrc/srx_gltf.cpp — scene ingestion: binary glTF (.glb) -> the flat arrays the
// packager and srx_scene_create take. Host code only. Replaces FileManager::loadGLTFFile / parseGLTFData
// (FileLoading.h:200-224, 438-782) together with what the reference does to the result before Scene::package sees it:
// Tri construction (Primitives.h:237-252), Object::fetchData / Packers::transformT flattening (Primitives.h:347-373,
// 423-434) with Object::compile_scale (:462-471) and Transformation::collpase (Transformation.h:28-57).
//
// The reference reads the file with tinygltf (a third-party header, not restated: this file has its own JSON and GLB
// reader) and decodes embedded images with stb_image; here images are PNG (own inflate; every depth, Adam7, tRNS) or Huffman-coded
// JPEG (own decoder following stb_image's arithmetic, checked against it), any other image fails loudly. Reference behaviour kept on purpose (each is what the author's scenes were exported for):
//  * axis swizzle (x, -z, y) for positions, normals and translations, (x, z, y) for scales, (x, -z, y, w) for
//    rotations (:450-452); uv -> (u, 1 - v) (:658-660); emissive x 25 (:473); sun emission = color * intensity / 50000
//    (:731), sun direction = (0, -1, 0) rotated (:726-727);
//  * only nodes that carry a camera, a mesh or a light exist (:687-742): transforms of plain group nodes are ignored;
//  * an accessor index of 0 for TEXCOORD_n / NORMAL means "absent" (:614,620: std::map::operator[] yields 0);
//  * a child's scale is child^2 * parent (Object::compile_scale multiplies its argument in twice);
//  * normals are rotated, never inverse-transpose scaled (Packers::transformT);
//  * KHR_materials_specular without specularColorFactor -> specular 0; without the extension -> 1 (:535-552);
//    IOR 1.5 / transmission 0 without their extensions (:553-574); a last "SR default material" (:579-587);
//  * textures: the loaders skip the first texel and read one past the end (Texture.h:41,57; quirk Q9; the last texel
//    is taken from a zero pad here), base colour goes through the luminance gamma 2.2 (converter_xyz_sRGB, :361-363)
//    and back to 8 bits by truncation (Texture.h:126-131), metallic / roughness are channels 2 / 1 re-quantised the
//    same way (Texture.h:214-226), normal maps are normalize((c*2-1)*(s,s,1)) stored as v*127+128 (Texture.h:51-67,
//    116-125); metallicFactor scales the first width*height BYTES of the metallic texture (:505-511).
// Differences: accessors honour byteOffset / byteStride as the glTF specification says (the reference copies the whole
// bufferView, :252-263 — identical for tightly packed, one-accessor views); children that are not mesh nodes are
// ignored (the reference indexes its object list with them, :743-755, undefined behaviour); no camera -> error.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/srx.h"

void srx_set_error(const std::string& s);

namespace {

struct Fail { std::string msg; };
[[noreturn]] void fail(const std::string& m) { throw Fail{m}; }

// ---------------------------------------------------------------- JSON
struct JV {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    double num = 0; bool b = false; std::string s;
    std::vector<JV> a;
    std::vector<std::pair<std::string, JV>> o;
    const JV* get(const char* k) const {
        if (t != OBJ) return nullptr;
        for (auto& kv : o) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const char* k) const { return get(k) != nullptr; }
    double number(const char* k, double def) const { const JV* v = get(k); return (v && v->t == NUM) ? v->num : def; }
    int integer(const char* k, int def) const { const JV* v = get(k); return (v && v->t == NUM) ? (int)v->num : def; }
    std::string str(const char* k) const { const JV* v = get(k); return (v && v->t == STR) ? v->s : std::string(); }
    size_t size() const { return t == ARR ? a.size() : 0; }
    const JV& at(size_t i) const { static const JV nul; return (t == ARR && i < a.size()) ? a[i] : nul; }
    double as_double() const { return t == NUM ? num : 0.0; }      // tinygltf Value::GetNumberAsDouble on a non-number: 0
};

struct JParser {
    const char* p; const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    JV parse() { ws(); JV v = value(0); ws(); return v; }
    JV value(int depth) {
        if (depth > 200) fail("glTF JSON nested too deep");
        ws();
        if (p >= e) fail("glTF JSON ends early");
        JV v;
        if (*p == '{') {
            v.t = JV::OBJ; p++; ws();
            if (p < e && *p == '}') { p++; return v; }
            for (;;) {
                ws();
                if (p >= e || *p != '"') fail("glTF JSON: expected a key");
                std::string k = string();
                ws();
                if (p >= e || *p != ':') fail("glTF JSON: expected ':'");
                p++;
                v.o.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; break; }
                fail("glTF JSON: expected ',' or '}'");
            }
        } else if (*p == '[') {
            v.t = JV::ARR; p++; ws();
            if (p < e && *p == ']') { p++; return v; }
            for (;;) {
                v.a.push_back(value(depth + 1));
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; break; }
                fail("glTF JSON: expected ',' or ']'");
            }
        } else if (*p == '"') {
            v.t = JV::STR; v.s = string();
        } else if (e - p >= 4 && !std::strncmp(p, "true", 4)) { v.t = JV::BOOL; v.b = true; p += 4; }
        else if (e - p >= 5 && !std::strncmp(p, "false", 5)) { v.t = JV::BOOL; v.b = false; p += 5; }
        else if (e - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; }
        else {
            std::string tok;
            while (p < e && (std::strchr("+-.eE", *p) || (*p >= '0' && *p <= '9'))) tok.push_back(*p++);
            if (tok.empty()) fail("glTF JSON: unexpected character");
            char* end = nullptr;
            v.t = JV::NUM; v.num = std::strtod(tok.c_str(), &end);
            if (!end || *end) fail("glTF JSON: bad number");
        }
        return v;
    }
    std::string string() {
        std::string s; p++;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) break;
                switch (*p) {
                case 'n': s.push_back('\n'); break; case 't': s.push_back('\t'); break; case 'r': s.push_back('\r'); break;
                case 'b': s.push_back('\b'); break; case 'f': s.push_back('\f'); break;
                case 'u': {
                    if (e - p < 5) fail("glTF JSON: bad \\u escape");
                    unsigned c = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                    p += 4;
                    if (c < 0x80) s.push_back((char)c);
                    else if (c < 0x800) { s.push_back((char)(0xC0 | (c >> 6))); s.push_back((char)(0x80 | (c & 0x3F))); }
                    else { s.push_back((char)(0xE0 | (c >> 12))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
                    break;
                }
                default: s.push_back(*p);
                }
                p++;
            } else s.push_back(*p++);
        }
        if (p >= e) fail("glTF JSON: unterminated string");
        p++;
        return s;
    }
};

// ---------------------------------------------------------------- reference vector arithmetic (XYZ.cpp), float as written
struct V3 { float X = 0, Y = 0, Z = 0; };
struct Q4 { float X = 0, Y = 0, Z = 0, W = 0; };
inline V3 add(V3 a, V3 b) { return {a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }
inline V3 sub(V3 a, V3 b) { return {a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }
inline V3 mul(V3 a, V3 b) { return {a.X * b.X, a.Y * b.Y, a.Z * b.Z}; }
inline V3 scale(V3 a, float s) { return {a.X * s, a.Y * s, a.Z * s}; }
inline V3 cross(V3 a, V3 b) { return {a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X}; }   // XYZ.cpp:292-298
inline V3 normalize(V3 v) { const float len = std::sqrt(v.X * v.X + v.Y * v.Y + v.Z * v.Z); return {v.X / len, v.Y / len, v.Z / len}; }   // :233-240
inline V3 apply_rotation(V3 p, Q4 r) {                                                    // XYZ.cpp:679-683
    const V3 rv{r.X, r.Y, r.Z};
    const V3 t = scale(cross(rv, p), 2.0f);
    return add(add(p, scale(t, r.W)), cross(rv, t));
}
inline Q4 q_multiply(Q4 q2, Q4 q1) {                                                      // XYZ.cpp:626-633
    return {q1.W * q2.X + q1.X * q2.W + q1.Y * q2.Z - q1.Z * q2.Y,
            q1.W * q2.Y - q1.X * q2.Z + q1.Y * q2.W + q1.Z * q2.X,
            q1.W * q2.Z + q1.X * q2.Y - q1.Y * q2.X + q1.Z * q2.W,
            q1.W * q2.W - q1.X * q2.X - q1.Y * q2.Y - q1.Z * q2.Z};
}
// XYZ(std::vector<double>, swizzle) with {0,-2,1} / Quat with {0,-2,1,3} / scale with {0,2,1} (XYZ.cpp:97,613)
inline V3 swz_pos(const JV& a) { return {(float)a.at(0).as_double(), (float)(-1 * a.at(2).as_double()), (float)(1 * a.at(1).as_double())}; }
inline V3 swz_scale(const JV& a) { return {(float)a.at(0).as_double(), (float)a.at(2).as_double(), (float)a.at(1).as_double()}; }
inline Q4 swz_rot(const JV& a) {
    return {(float)a.at(0).as_double(), (float)(-1 * a.at(2).as_double()), (float)(1 * a.at(1).as_double()), (float)(1 * a.at(3).as_double())};
}
inline V3 swz(V3 v) { return {1 * v.X, -1 * v.Z, 1 * v.Y}; }                              // XYZ::swizzle, XYZ.cpp:115-121

struct Transform { bool has_rot = false, has_move = false; Q4 rot; V3 move; };              // Transformation.h:6-27
inline void stack(const Transform& t, V3& pos, Q4& rot) {                                   // Transformation.h:28-36
    if (t.has_move) pos = add(pos, t.move);
    if (t.has_rot) { rot = q_multiply(rot, t.rot); pos = apply_rotation(pos, t.rot); }
}

struct Tri { V3 p1, p2, p3; float uv[6]; V3 n1, n2, n3; int material; };
inline Tri make_tri(V3 p1, V3 p2, V3 p3, const float* uv, int material) {                   // Primitives.h:237-252
    Tri t; t.p1 = p1; t.p2 = p2; t.p3 = p3; std::memcpy(t.uv, uv, sizeof(t.uv)); t.material = material;
    const V3 normal = normalize(cross(sub(p3, p1), sub(p2, p1)));
    t.n1 = t.n2 = t.n3 = normal;
    return t;
}

struct Object {
    int parent = -1;
    V3 origin, scale{1, 1, 1};
    std::vector<Transform> transformations;
    std::vector<int> children;
    int mesh = -1;
};

// ---------------------------------------------------------------- inflate + PNG (images inside the .glb)
struct BitReader {
    const uint8_t* d; size_t n, pos = 0; uint32_t buf = 0; int cnt = 0;
    uint32_t bits(int k) {
        while (cnt < k) { if (pos >= n) fail("PNG: compressed data ends early"); buf |= (uint32_t)d[pos++] << cnt; cnt += 8; }
        const uint32_t v = buf & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        buf >>= k; cnt -= k;
        return v;
    }
};
struct Huff {
    uint16_t count[16] = {0}; uint16_t symbol[320] = {0};
    void build(const uint8_t* len, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)br.bits(1);
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        fail("PNG: bad Huffman code");
    }
};
// max_out: what the caller can use (an image's (stride + 1) x height); a stream that inflates to more is refused, so a
// small file cannot exhaust memory
std::vector<uint8_t> inflate_zlib(const uint8_t* d, size_t n, size_t max_out) {
    if (n < 6) fail("PNG: zlib stream too short");
    BitReader br{d + 2, n - 2};
    std::vector<uint8_t> out;
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (bool last = false; !last;) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (type == 0) {
            br.buf = 0; br.cnt = 0;
            if (br.pos + 4 > br.n) fail("PNG: stored block header cut off");
            const uint32_t len = br.d[br.pos] | (br.d[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) fail("PNG: stored block cut off");
            if (out.size() + len > max_out) fail("PNG: the compressed stream holds more data than the image needs");
            out.insert(out.end(), br.d + br.pos, br.d + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) fail("PNG: bad deflate block type");
        Huff lit, dist;
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            lit.build(lens, 288);
            for (int i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
            Huff h; h.build(cl, 19);
            int i = 0;
            while (i < nlen + ndist) {
                const int sym = h.decode(br);
                if (sym < 16) lens[i++] = (uint8_t)sym;
                else {
                    int rep; uint8_t v = 0;
                    if (sym == 16) { if (i == 0) fail("PNG: bad code lengths"); v = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (i + rep > nlen + ndist) fail("PNG: bad code lengths");
                    while (rep--) lens[i++] = v;
                }
            }
            lit.build(lens, nlen);
            dist.build(lens + nlen, ndist);
        }
        for (;;) {
            const int sym = lit.decode(br);
            if (sym < 256) { if (out.size() >= max_out) fail("PNG: the compressed stream holds more data than the image needs"); out.push_back((uint8_t)sym); }
            else if (sym == 256) break;
            else {
                if (sym > 285) fail("PNG: bad length symbol");
                const int len = lbase[sym - 257] + (int)br.bits(lext[sym - 257]);
                const int ds = dist.decode(br);
                if (ds > 29) fail("PNG: bad distance symbol");
                const size_t dd = dbase[ds] + br.bits(dext[ds]);
                if (dd > out.size()) fail("PNG: distance beyond the start");
                if (out.size() + (size_t)len > max_out) fail("PNG: the compressed stream holds more data than the image needs");
                for (int k = 0; k < len; k++) out.push_back(out[out.size() - dd]);
            }
        }
    }
    return out;
}

struct Image { int w = 0, h = 0, comp = 0; std::vector<uint8_t> px; };   // comp 3 or 4, rows top to bottom (stb_image layout)
// PNG as stb_image reads it with 4 components requested (stb_image.h:4600-5200): colour types 0 / 2 / 3 / 4 / 6, bit
// depths 1 / 2 / 4 / 8 / 16 (gray below 8 bits is scaled to 0-255: x 255 / 85 / 17; 16-bit samples keep their HIGH byte),
// Adam7 interlacing (seven sub-images, each filtered on its own), tRNS (palette alphas; a colour key for gray and RGB,
// compared at the file's depth). Ancillary chunks are ignored.
Image decode_png(const uint8_t* d, size_t n) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || std::memcmp(d, sig, 8)) fail("image is neither a PNG nor a JPEG (PNG and Huffman-coded JPEG images are decoded; the reference uses stb_image)");
    Image im; int ctype = -1, depth = 0, interlace = 0;
    std::vector<uint8_t> z, plte, trns;
    for (size_t at = 8; at + 12 <= n;) {
        const uint32_t len = (uint32_t)d[at] << 24 | d[at + 1] << 16 | d[at + 2] << 8 | d[at + 3];
        const char* ty = (const char*)d + at + 4;
        const uint8_t* body = d + at + 8;
        if (at + 12 + (size_t)len > n) fail("PNG: chunk cut off");
        if (!std::strncmp(ty, "IHDR", 4)) {
            if (len < 13) fail("PNG: bad IHDR");
            im.w = (int)((uint32_t)body[0] << 24 | body[1] << 16 | body[2] << 8 | body[3]);
            im.h = (int)((uint32_t)body[4] << 24 | body[5] << 16 | body[6] << 8 | body[7]);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) fail("PNG: bit depth must be 1, 2, 4, 8 or 16");
            if (!(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6)) fail("PNG: bad colour type");
            if ((ctype == 3 && depth == 16) || ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8)) fail("PNG: bit depth not allowed for this colour type");
            if (body[10] != 0 || body[11] != 0 || interlace > 1) fail("PNG: bad compression, filter or interlace method");
        } else if (!std::strncmp(ty, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::strncmp(ty, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::strncmp(ty, "IDAT", 4)) z.insert(z.end(), body, body + len);
        else if (!std::strncmp(ty, "IEND", 4)) break;
        at += 12 + (size_t)len;
    }
    if (im.w <= 0 || im.h <= 0 || ctype < 0) fail("PNG: no IHDR");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    if ((uint64_t)im.w * (uint64_t)im.h > (1ull << 28)) fail("PNG: image larger than 2^28 pixels");
    // the sub-images: one, or Adam7's seven
    static const int xorig[7] = {0, 4, 0, 2, 0, 1, 0}, yorig[7] = {0, 0, 4, 0, 2, 0, 1}, xspc[7] = {8, 8, 4, 4, 2, 2, 1}, yspc[7] = {8, 8, 8, 4, 4, 2, 2};
    struct Pass { int w, h, x0, y0, dx, dy; size_t row_bytes; };
    std::vector<Pass> passes;
    size_t raw_need = 0;
    for (int p = 0; p < (interlace ? 7 : 1); p++) {
        Pass q;
        if (interlace) { q.x0 = xorig[p]; q.y0 = yorig[p]; q.dx = xspc[p]; q.dy = yspc[p]; q.w = (im.w - q.x0 + q.dx - 1) / q.dx; q.h = (im.h - q.y0 + q.dy - 1) / q.dy; }
        else { q.x0 = q.y0 = 0; q.dx = q.dy = 1; q.w = im.w; q.h = im.h; }
        if (q.w <= 0 || q.h <= 0) continue;
        q.row_bytes = ((size_t)ch * q.w * depth + 7) >> 3;
        raw_need += (q.row_bytes + 1) * (size_t)q.h;
        passes.push_back(q);
    }
    std::vector<uint8_t> raw = inflate_zlib(z.data(), z.size(), raw_need);
    if (raw.size() < raw_need) fail("PNG: pixel data too short");
    // samples at the file's depth (16-bit ones whole), `ch` per pixel
    std::vector<uint16_t> smp((size_t)im.w * im.h * ch);
    const int fb = depth == 16 ? 2 * ch : (depth == 8 ? ch : 1);      // bytes a filter looks back
    size_t at = 0;
    std::vector<uint8_t> cur_row, up_row;
    for (const Pass& q : passes) {
        cur_row.assign(q.row_bytes, 0); up_row.assign(q.row_bytes, 0);
        for (int y = 0; y < q.h; y++) {
            const uint8_t* in = raw.data() + at;
            at += q.row_bytes + 1;
            const int f = in[0];
            if (f > 4) fail("PNG: bad filter");
            for (size_t x = 0; x < q.row_bytes; x++) {
                const int a = x >= (size_t)fb ? cur_row[x - fb] : 0, b2 = y ? up_row[x] : 0, c = (y && x >= (size_t)fb) ? up_row[x - fb] : 0;
                int v = in[1 + x];
                switch (f) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b2; break;
                case 3: v += (a + b2) >> 1; break;
                default: { const int pp = a + b2 - c, pa = std::abs(pp - a), pb = std::abs(pp - b2), pc = std::abs(pp - c);
                           v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b2 : c); break; }
                }
                cur_row[x] = (uint8_t)v;
            }
            const int oy = q.y0 + y * q.dy;
            for (int x = 0; x < q.w; x++) {
                uint16_t* o = smp.data() + ((size_t)oy * im.w + (size_t)(q.x0 + x * q.dx)) * ch;
                for (int k = 0; k < ch; k++) {
                    const size_t i = (size_t)x * ch + k;
                    if (depth == 16) o[k] = (uint16_t)((cur_row[2 * i] << 8) | cur_row[2 * i + 1]);
                    else if (depth == 8) o[k] = cur_row[i];
                    else { const size_t bit = i * depth; o[k] = (uint16_t)((cur_row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                }
            }
            cur_row.swap(up_row);
        }
    }
    // to RGBA, 8 bits (tinygltf asks stb_image for 4 components)
    const int gray_scale = depth == 1 ? 0xff : depth == 2 ? 0x55 : depth == 4 ? 0x11 : 1;      // stbi__depth_scale_table
    bool has_key = false; uint16_t key[3] = {0, 0, 0};
    if (!trns.empty() && (ctype == 0 || ctype == 2)) {
        if (trns.size() != (size_t)(ctype == 0 ? 2 : 6)) fail("PNG: bad tRNS length");
        has_key = true;
        for (int k = 0; k < (ctype == 0 ? 1 : 3); k++) {
            const uint16_t v = (uint16_t)((trns[2 * k] << 8) | trns[2 * k + 1]);
            key[k] = depth == 16 ? v : (uint16_t)((v & 255) * gray_scale);                  // the 8-bit key is compared with the SCALED samples
        }
    }
    if (ctype == 3 && (plte.empty() || plte.size() % 3 || plte.size() > 768)) fail("PNG: bad or missing palette");
    if (ctype == 3 && trns.size() > plte.size() / 3) fail("PNG: tRNS longer than the palette");
    im.comp = 4;
    im.px.resize((size_t)im.w * im.h * 4);
    auto to8 = [&](uint16_t v) { return (uint8_t)(depth == 16 ? (v >> 8) : v); };
    for (size_t i = 0; i < (size_t)im.w * im.h; i++) {
        const uint16_t* sp = smp.data() + i * ch;
        uint8_t r, g, b, a = 255;
        if (ctype == 3) {
            const size_t idx = sp[0];
            if (idx * 3 + 2 >= plte.size()) fail("PNG: palette index out of range");
            r = plte[idx * 3]; g = plte[idx * 3 + 1]; b = plte[idx * 3 + 2];
            if (idx < trns.size()) a = trns[idx];
        } else if (ctype == 0 || ctype == 4) {
            const uint16_t v = depth < 8 ? (uint16_t)(sp[0] * gray_scale) : sp[0];
            r = g = b = to8(v);
            if (ctype == 4) a = to8(sp[1]);
            else if (has_key && v == key[0]) a = 0;
        } else {
            r = to8(sp[0]); g = to8(sp[1]); b = to8(sp[2]);
            if (ctype == 6) a = to8(sp[3]);
            else if (has_key && sp[0] == key[0] && sp[1] == key[1] && sp[2] == key[2]) a = 0;
        }
        im.px[4 * i] = r; im.px[4 * i + 1] = g; im.px[4 * i + 2] = b; im.px[4 * i + 3] = a;
    }
    return im;
}

// ---------------------------------------------------------------- JPEG (baseline / extended sequential / progressive Huffman, 8 bits)
// The reference decodes images with stb_image 2.28 (through tinygltf, 4 components requested). Texel values feed the
// shading, so the decoder follows stb_image's arithmetic step for step: the integer slow IDCT of jidctint with 12-bit
// constants and stb's two extra bits between the passes, its chroma upsampling filters (h2v1: 3/4 + 1/4 across, h1v2:
// the same down, h2v2: both, anything else: nearest) and its 20-bit fixed-point YCbCr -> RGB. tests/test_gltf_cpu.py
// compares the decoded bytes with stb_image's own (oracle/_ref/ref_stb_probe, built from the reference's vendored header).
// Progressive files (SOF2) are decoded like stb_image does it: coefficients accumulated over the scans, dequantised and
// transformed at the end. Arithmetic-coded files, 12-bit samples and CMYK fail loudly.
struct JpegHuff { uint8_t size[257]; uint16_t code[256]; uint8_t values[256]; int maxcode[18]; int delta[17]; int n = 0; bool set = false; };
struct JpegComp { int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0; int x = 0, y = 0, w2 = 0, h2 = 0; std::vector<uint8_t> data;
                  std::vector<short> coeff; int coeff_w = 0; };     // progressive: the coefficients of every block, kept across the scans
struct JpegDec {
    const uint8_t* d; size_t n, pos = 0;
    uint32_t code_buffer = 0; int code_bits = 0; int marker = -1; bool nomore = false;
    JpegHuff hdc[4], hac[4];
    uint16_t dequant[4][64];
    JpegComp comp[4];
    int ncomp = 0, img_x = 0, img_y = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0, mcu_w = 0, mcu_h = 0;
    int restart_interval = 0, todo = 0;
    int scan_n = 0, order[4];
    bool jfif = false; int app14 = -1; int rgb = 0;
    // progressive (SOF2): spectral selection and successive approximation of the current scan, pending end-of-band run
    bool progressive = false; int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;

    int get8() { return pos < n ? d[pos++] : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }
    void build(JpegHuff& h, const int* count) {                      // ITU T.81 Annex C, as every decoder does
        int k = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < count[i]; j++) { h.size[k++] = (uint8_t)(i + 1); if (k >= 257) fail("JPEG: bad code lengths"); }
        h.size[k] = 0; h.n = k;
        int code = 0; k = 0;
        for (int j = 1; j <= 16; j++) {
            h.delta[j] = k - code;
            if (h.size[k] == j) {
                while (h.size[k] == j) h.code[k++] = (uint16_t)(code++);
                if (code - 1 >= (1 << j)) fail("JPEG: bad code lengths");
            }
            h.maxcode[j] = code << (16 - j);
            code <<= 1;
        }
        h.maxcode[17] = 0x7fffffff;
        h.set = true;
    }
    void grow() {                                                     // bytes into the bit buffer; a marker ends the entropy-coded data
        do {
            unsigned b = nomore ? 0 : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();
                if (c != 0) { marker = c; nomore = true; return; }
            }
            code_buffer |= b << (24 - code_bits);
            code_bits += 8;
        } while (code_bits <= 24);
    }
    int huff(const JpegHuff& h) {
        if (code_bits < 16) grow();
        const unsigned temp = code_buffer >> 16;
        int k = 1;
        while (k <= 16 && (int)temp >= h.maxcode[k]) k++;
        if (k == 17) { code_bits -= 16; fail("JPEG: bad huffman code"); }
        if (k > code_bits) fail("JPEG: bad huffman code");
        const int c = (int)((code_buffer >> (32 - k)) & ((1u << k) - 1u)) + h.delta[k];
        if (c < 0 || c >= h.n) fail("JPEG: bad huffman code");
        code_bits -= k; code_buffer <<= k;
        return h.values[c];
    }
    int extend_receive(int nb) {                                      // T.81 F.2.2.1: nb bits, sign-extended
        if (nb == 0) return 0;
        if (code_bits < nb) grow();
        if (code_bits < nb) return 0;
        const unsigned v = code_buffer >> (32 - nb);
        code_buffer <<= nb; code_bits -= nb;
        return (int)v < (1 << (nb - 1)) ? (int)v - (1 << nb) + 1 : (int)v;
    }
    int get_bit() {                                                   // stb_image's stbi__jpeg_get_bit: 0 once the stream has run out
        if (code_bits < 1) grow();
        if (code_bits < 1) return 0;
        const unsigned k = code_buffer;
        code_buffer <<= 1; --code_bits;
        return (int)(k >> 31);
    }
    int get_bits(int nb) {
        if (nb == 0) return 0;
        if (code_bits < nb) grow();
        if (code_bits < nb) return 0;
        const unsigned v = code_buffer >> (32 - nb);
        code_buffer <<= nb; code_bits -= nb;
        return (int)v;
    }
    // Progressive blocks (T.81 Annex G), in stb_image's order of operations (stb_image.h:2265-2397): a DC scan sets or
    // refines data[0]; an AC scan fills or refines the band spec_start..spec_end, with end-of-band runs across blocks.
    // Coefficients stay UNdequantised until the end (finish()).
    void decode_block_prog_dc(short data[64], JpegComp& c) {
        if (spec_end != 0) fail("JPEG: a progressive scan cannot carry DC and AC coefficients together");
        if (code_bits < 16) grow();
        if (succ_high == 0) {
            std::memset(data, 0, 64 * sizeof(short));
            const int t = huff(hdc[c.hd]);
            if (t > 15) fail("JPEG: bad DC code");
            const int diff = t ? extend_receive(t) : 0;
            const long long dc = (long long)c.dc_pred + diff;
            if (dc > 32767 || dc < -32768 || dc * (1LL << succ_low) > 32767 || dc * (1LL << succ_low) < -32768) fail("JPEG: DC coefficient out of range");
            c.dc_pred = (int)dc;
            data[0] = (short)(dc * (1 << succ_low));
        } else if (get_bit()) {
            data[0] = (short)(data[0] + (short)(1 << succ_low));
        }
    }
    void decode_block_prog_ac(short data[64], JpegComp& c) {
        static const uint8_t dezigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
            63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
        if (spec_start == 0) fail("JPEG: a progressive scan cannot carry DC and AC coefficients together");
        const JpegHuff& h = hac[c.ha];
        if (succ_high == 0) {
            const int shift = succ_low;
            if (eob_run) { --eob_run; return; }
            int k = spec_start;
            do {
                const int rs = huff(h);
                const int sz = rs & 15, r = rs >> 4;
                if (sz == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int zig = dezigzag[k++];
                    data[zig] = (short)((uint32_t)extend_receive(sz) * (1u << shift));
                }
            } while (k <= spec_end);
        } else {
            const short bit = (short)(1 << succ_low);
            auto refine = [&](short* p) {
                if (get_bit() && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) {
                    short* p = &data[dezigzag[k]];
                    if (*p != 0) refine(p);
                }
            } else {
                int k = spec_start;
                do {
                    const int rs = huff(h);
                    int sz = rs & 15, r = rs >> 4;
                    if (sz == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += get_bits(r);
                            r = 64;                                   // force the end of the block
                        }                                             // (r = 15: a run of 16 zeros, nothing special to do)
                    } else {
                        if (sz != 1) fail("JPEG: bad huffman code");
                        sz = get_bit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        short* p = &data[dezigzag[k++]];
                        if (*p != 0) refine(p);
                        else {
                            if (r == 0) { *p = (short)sz; break; }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
    }
    // progressive: after the last scan, dequantise every block (in 16-bit arithmetic, as stb_image does) and transform it
    void finish() {
        for (int n2 = 0; n2 < ncomp; n2++) {
            JpegComp& c = comp[n2];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            const uint16_t* dq = dequant[c.tq];
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    short* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int k = 0; k < 64; k++) data[k] = (short)((uint32_t)(int)data[k] * dq[k]);
                    idct(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
        }
    }
    void decode_block(short data[64], JpegComp& c) {
        static const uint8_t dezigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
            63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
        const uint16_t* dq = dequant[c.tq];
        const int t = huff(hdc[c.hd]);
        if (t > 15) fail("JPEG: bad DC code");
        std::memset(data, 0, 64 * sizeof(short));
        const int diff = t ? extend_receive(t) : 0;
        const int dc = (int)((uint32_t)c.dc_pred + (uint32_t)diff);
        c.dc_pred = dc;
        data[0] = (short)((uint32_t)dc * dq[0]);
        for (int k = 1; k < 64;) {
            const int rs = huff(hac[c.ha]);
            const int sz = rs & 15, r = rs >> 4;
            if (sz == 0) {
                if (rs != 0xf0) break;                                // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63) fail("JPEG: coefficient index out of range");
                const int zig = dezigzag[k++];
                data[zig] = (short)((uint32_t)extend_receive(sz) * dq[zig]);
            }
        }
    }
    static uint8_t clamp8(int x) { return (unsigned)x > 255 ? (x < 0 ? 0 : 255) : (uint8_t)x; }
    // jidctint's slow integer IDCT with 12-bit constants, two extra bits kept between the passes (stb_image's form of it)
    static void idct(uint8_t* out, int stride, const short data[64]) {
        auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
        static const int c0541 = f2f(0.5411961f), c1847 = f2f(-1.847759065f), c0765 = f2f(0.765366865f), c1175 = f2f(1.175875602f),
                         c0298 = f2f(0.298631336f), c2053 = f2f(2.053119869f), c3072 = f2f(3.072711026f), c1501 = f2f(1.501321110f),
                         c0899 = f2f(-0.899976223f), c2562 = f2f(-2.562915447f), c1961 = f2f(-1.961570560f), c0390 = f2f(-0.390180644f);
        // 32-bit wrap-around arithmetic (unsigned) so that damaged files cannot overflow a signed int; for valid data it is
        // the same as stb_image's int arithmetic
        typedef uint32_t U;
        auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, U& x0, U& x1, U& x2, U& x3, U& t0, U& t1, U& t2, U& t3) {
            U p2 = (U)s2, p3 = (U)s6;
            U p1 = (p2 + p3) * (U)c0541;
            t2 = p1 + p3 * (U)c1847; t3 = p1 + p2 * (U)c0765;
            p2 = (U)s0; p3 = (U)s4;
            t0 = (p2 + p3) * 4096u; t1 = (p2 - p3) * 4096u;
            x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
            t0 = (U)s7; t1 = (U)s5; t2 = (U)s3; t3 = (U)s1;
            p3 = t0 + t2; U p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
            const U p5 = (p3 + p4) * (U)c1175;
            t0 = t0 * (U)c0298; t1 = t1 * (U)c2053; t2 = t2 * (U)c3072; t3 = t3 * (U)c1501;
            p1 = p5 + p1 * (U)c0899; p2 = p5 + p2 * (U)c2562; p3 = p3 * (U)c1961; p4 = p4 * (U)c0390;
            t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        };
        auto sar = [](U x, int k) { return (int)x >> k; };
        int val[64];
        for (int i = 0; i < 8; i++) {
            const short* dd = data + i; int* v = val + i;
            if (dd[8] == 0 && dd[16] == 0 && dd[24] == 0 && dd[32] == 0 && dd[40] == 0 && dd[48] == 0 && dd[56] == 0) {
                const int dcterm = dd[0] * 4;                         // a short times 4: no overflow
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
            } else {
                U x0, x1, x2, x3, t0, t1, t2, t3;
                pass(dd[0], dd[8], dd[16], dd[24], dd[32], dd[40], dd[48], dd[56], x0, x1, x2, x3, t0, t1, t2, t3);
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = sar(x0 + t3, 10); v[56] = sar(x0 - t3, 10); v[8] = sar(x1 + t2, 10); v[48] = sar(x1 - t2, 10);
                v[16] = sar(x2 + t1, 10); v[40] = sar(x2 - t1, 10); v[24] = sar(x3 + t0, 10); v[32] = sar(x3 - t0, 10);
            }
        }
        for (int i = 0; i < 8; i++) {
            const int* v = val + 8 * i; uint8_t* o = out + (size_t)stride * i;
            U x0, x1, x2, x3, t0, t1, t2, t3;
            pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
            const U bias = 65536u + (128u << 17);
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            o[0] = clamp8(sar(x0 + t3, 17)); o[7] = clamp8(sar(x0 - t3, 17)); o[1] = clamp8(sar(x1 + t2, 17)); o[6] = clamp8(sar(x1 - t2, 17));
            o[2] = clamp8(sar(x2 + t1, 17)); o[5] = clamp8(sar(x2 - t1, 17)); o[3] = clamp8(sar(x3 + t0, 17)); o[4] = clamp8(sar(x3 - t0, 17));
        }
    }
    void reset_entropy() {
        code_bits = 0; code_buffer = 0; nomore = false; marker = -1; eob_run = 0;
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
    }
    void scan() {
        reset_entropy();
        short data[64];
        auto restart = [&]() {                                        // after every restart interval: an RSTn marker, then a fresh decoder state
            if (--todo <= 0) {
                if (code_bits < 24) grow();
                if (!(marker >= 0xd0 && marker <= 0xd7)) return false;
                reset_entropy();
            }
            return true;
        };
        if (progressive) {                                            // stb_image.h:3010-3070
            if (scan_n == 1) {
                JpegComp& c = comp[order[0]];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                if (spec_start == 0 ? !hdc[c.hd].set : !hac[c.ha].set) fail("JPEG: scan uses a Huffman table that was not defined");
                for (int j = 0; j < h; j++)
                    for (int i = 0; i < w; i++) {
                        short* blk = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                        if (spec_start == 0) decode_block_prog_dc(blk, c); else decode_block_prog_ac(blk, c);
                        if (!restart()) return;
                    }
            } else {                                                  // interleaved: DC only
                for (int j = 0; j < mcu_y; j++)
                    for (int i = 0; i < mcu_x; i++) {
                        for (int k = 0; k < scan_n; k++) {
                            JpegComp& c = comp[order[k]];
                            if (!hdc[c.hd].set) fail("JPEG: scan uses a Huffman table that was not defined");
                            for (int y = 0; y < c.v; y++)
                                for (int x = 0; x < c.h; x++) {
                                    const int x2 = i * c.h + x, y2 = j * c.v + y;
                                    decode_block_prog_dc(c.coeff.data() + 64 * ((size_t)x2 + (size_t)y2 * c.coeff_w), c);
                                }
                        }
                        if (!restart()) return;
                    }
            }
            return;
        }
        if (scan_n == 1) {                                            // one component: its own block grid, no MCU padding
            JpegComp& c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    if (!hdc[c.hd].set || !hac[c.ha].set) fail("JPEG: scan uses a Huffman table that was not defined");
                    decode_block(data, c);
                    idct(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                    if (!restart()) return;
                }
        } else {
            for (int j = 0; j < mcu_y; j++)
                for (int i = 0; i < mcu_x; i++) {
                    for (int k = 0; k < scan_n; k++) {
                        JpegComp& c = comp[order[k]];
                        if (!hdc[c.hd].set || !hac[c.ha].set) fail("JPEG: scan uses a Huffman table that was not defined");
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                                decode_block(data, c);
                                idct(c.data.data() + (size_t)c.w2 * y2 + x2, c.w2, data);
                            }
                    }
                    if (!restart()) return;
                }
        }
    }
    void frame_header(int m) {
        progressive = m == 0xc2;
        const int lf = get16();
        if (lf < 11) fail("JPEG: bad SOF length");
        if (get8() != 8) fail("JPEG: only 8-bit samples are decoded");
        img_y = get16(); img_x = get16();
        if (img_x <= 0 || img_y <= 0) fail("JPEG: bad size");
        if ((uint64_t)img_x * (uint64_t)img_y > (1ull << 28)) fail("JPEG: image larger than 2^28 pixels");
        ncomp = get8();
        if (ncomp == 4) fail("JPEG: CMYK / 4-component files are not decoded");
        if (ncomp != 1 && ncomp != 3) fail("JPEG: bad component count");
        if (lf != 8 + 3 * ncomp) fail("JPEG: bad SOF length");
        rgb = 0;
        for (int i = 0; i < ncomp; i++) {
            static const uint8_t rgbid[3] = {'R', 'G', 'B'};
            comp[i].id = get8();
            if (ncomp == 3 && comp[i].id == rgbid[i]) rgb++;
            const int q = get8();
            comp[i].h = q >> 4; comp[i].v = q & 15;
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) fail("JPEG: bad sampling factors");
            comp[i].tq = get8();
            if (comp[i].tq > 3) fail("JPEG: bad quantisation table index");
        }
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; i++) { h_max = std::max(h_max, comp[i].h); v_max = std::max(v_max, comp[i].v); }
        for (int i = 0; i < ncomp; i++) if (h_max % comp[i].h || v_max % comp[i].v) fail("JPEG: sampling factors that do not divide the largest are not decoded");
        mcu_w = h_max * 8; mcu_h = v_max * 8;
        mcu_x = (img_x + mcu_w - 1) / mcu_w; mcu_y = (img_y + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; i++) {
            comp[i].x = (img_x * comp[i].h + h_max - 1) / h_max;
            comp[i].y = (img_y * comp[i].v + v_max - 1) / v_max;
            comp[i].w2 = mcu_x * comp[i].h * 8; comp[i].h2 = mcu_y * comp[i].v * 8;
            comp[i].data.assign((size_t)comp[i].w2 * comp[i].h2 + 15, 0);
            if (progressive) { comp[i].coeff_w = comp[i].w2 / 8; comp[i].coeff.assign((size_t)comp[i].w2 * comp[i].h2 + 15, 0); }
        }
    }
    void scan_header() {
        const int ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > ncomp) fail("JPEG: bad SOS component count");
        if (ls != 6 + 2 * scan_n) fail("JPEG: bad SOS length");
        for (int i = 0; i < scan_n; i++) {
            const int id = get8(), q = get8();
            int which = 0;
            for (; which < ncomp; which++) if (comp[which].id == id) break;
            if (which == ncomp) fail("JPEG: scan names a component the frame does not have");
            comp[which].hd = q >> 4; comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) fail("JPEG: bad Huffman table index");
            order[i] = which;
        }
        spec_start = get8(); spec_end = get8();
        const int a = get8();
        succ_high = a >> 4; succ_low = a & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) fail("JPEG: bad progressive scan parameters");
        } else {
            if (spec_start != 0 || a != 0) fail("JPEG: not a baseline scan");
            spec_end = 63;
        }
    }
    void marker_segment(int m) {
        if (m == 0xdd) { if (get16() != 4) fail("JPEG: bad DRI length"); restart_interval = get16(); return; }
        if (m == 0xdb) {                                              // quantisation tables, stored de-zigzagged
            static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
            int l = get16() - 2;
            while (l > 0) {
                const int q = get8(), p = q >> 4, t = q & 15;
                if ((p != 0 && p != 1) || t > 3) fail("JPEG: bad DQT");
                for (int i = 0; i < 64; i++) dequant[t][zz[i]] = (uint16_t)(p ? get16() : get8());
                l -= p ? 129 : 65;
            }
            if (l != 0) fail("JPEG: bad DQT length");
            return;
        }
        if (m == 0xc4) {                                              // Huffman tables
            int l = get16() - 2;
            while (l > 0) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) fail("JPEG: bad DHT");
                int count[16], total = 0;
                for (int i = 0; i < 16; i++) { count[i] = get8(); total += count[i]; }
                if (total > 256) fail("JPEG: bad DHT");
                JpegHuff& h = tc ? hac[th] : hdc[th];
                build(h, count);
                for (int i = 0; i < total; i++) h.values[i] = (uint8_t)get8();
                l -= 17 + total;
            }
            if (l != 0) fail("JPEG: bad DHT length");
            return;
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {                  // APPn / COM: JFIF and Adobe say how the components are to be read
            int l = get16();
            if (l < 2) fail("JPEG: bad segment length");
            l -= 2;
            if (m == 0xe0 && l >= 5) {
                static const char tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; i++) if (get8() != (uint8_t)tag[i]) ok = false;
                l -= 5;
                if (ok) jfif = true;
            } else if (m == 0xee && l >= 12) {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; i++) if (get8() != (uint8_t)tag[i]) ok = false;
                l -= 6;
                if (ok) { get8(); get16(); get16(); app14 = get8(); l -= 6; }
            }
            if (pos + (size_t)l > n) fail("JPEG: segment cut off");
            pos += (size_t)l;
            return;
        }
        fail("JPEG: unknown marker");
    }
    int next_marker() {
        if (marker >= 0) { const int m = marker; marker = -1; return m; }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }
};

Image decode_jpeg(const uint8_t* d, size_t n) {
    std::unique_ptr<JpegDec> zp(new JpegDec{d, n});
    JpegDec& z = *zp;
    std::memset(z.dequant, 0, sizeof(z.dequant));
    if (z.get8() != 0xff || z.get8() != 0xd8) fail("JPEG: no SOI");
    int m = z.next_marker();
    while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
        if (m < 0 || z.pos >= n) fail("JPEG: no frame header");
        if (m == 0xc9 || m == 0xca || m == 0xcb || m == 0xc3) fail("JPEG: arithmetic-coded and lossless files are not decoded");
        z.marker_segment(m);
        m = z.next_marker();
    }
    z.frame_header(m);
    bool seen_scan = false;
    for (m = z.next_marker(); m != 0xd9; m = z.next_marker()) {
        if (m == 0xda) {
            z.scan_header();
            z.scan();
            seen_scan = true;
            if (z.marker < 0) {                                       // skip whatever follows the entropy-coded data up to the next marker
                while (z.pos < n) { if (z.get8() == 0xff) { const int c = z.get8(); if (c != 0 && c != 0xff) { z.marker = c; break; } } }
                if (z.marker < 0) break;
            }
        } else if (m == 0xdc) { z.get16(); z.get16(); }               // DNL: the height was in the frame header
        else if (m < 0) { if (z.pos >= n) break; fail("JPEG: data where a marker should be"); }
        else z.marker_segment(m);
    }
    if (!seen_scan) fail("JPEG: no scan");
    if (z.progressive) z.finish();
    // ---- upsample and convert, row by row (stb_image's load_jpeg_image with 4 components requested)
    Image im; im.w = z.img_x; im.h = z.img_y; im.comp = 4;
    im.px.resize((size_t)im.w * im.h * 4);
    const bool is_rgb = z.ncomp == 3 && (z.rgb == 3 || (z.app14 == 0 && !z.jfif));
    struct Res { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
    Res res[3];
    for (int k = 0; k < z.ncomp; k++) {
        Res& r = res[k];
        r.hs = z.h_max / z.comp[k].h; r.vs = z.v_max / z.comp[k].v;
        r.ystep = r.vs >> 1; r.w_lores = (z.img_x + r.hs - 1) / r.hs; r.ypos = 0;
        r.line0 = r.line1 = z.comp[k].data.data();
        r.buf.assign((size_t)z.img_x + 3 + 8, 0);
    }
    auto div4 = [](int x) { return (uint8_t)(x >> 2); };
    auto div16 = [](int x) { return (uint8_t)(x >> 4); };
    for (int j = 0; j < z.img_y; j++) {
        const uint8_t* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < z.ncomp; k++) {
            Res& r = res[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            const uint8_t* near = y_bot ? r.line1 : r.line0;
            const uint8_t* far = y_bot ? r.line0 : r.line1;
            uint8_t* out = r.buf.data();
            const int w = r.w_lores;
            if (r.hs == 1 && r.vs == 1) row[k] = near;
            else {
                if (r.hs == 1 && r.vs == 2) { for (int i = 0; i < w; i++) out[i] = div4(3 * near[i] + far[i] + 2); }
                else if (r.hs == 2 && r.vs == 1) {
                    if (w == 1) out[0] = out[1] = near[0];
                    else {
                        out[0] = near[0]; out[1] = div4(near[0] * 3 + near[1] + 2);
                        int i = 1;
                        for (; i < w - 1; i++) { const int t = 3 * near[i] + 2; out[i * 2] = div4(t + near[i - 1]); out[i * 2 + 1] = div4(t + near[i + 1]); }
                        out[i * 2] = div4(near[w - 2] * 3 + near[w - 1] + 2); out[i * 2 + 1] = near[w - 1];
                    }
                } else if (r.hs == 2 && r.vs == 2) {
                    if (w == 1) out[0] = out[1] = div4(3 * near[0] + far[0] + 2);
                    else {
                        int t1 = 3 * near[0] + far[0];
                        out[0] = div4(t1 + 2);
                        for (int i = 1; i < w; i++) { const int t0 = t1; t1 = 3 * near[i] + far[i]; out[i * 2 - 1] = div16(3 * t0 + t1 + 8); out[i * 2] = div16(3 * t1 + t0 + 8); }
                        out[w * 2 - 1] = div4(t1 + 2);
                    }
                } else { for (int i = 0; i < w; i++) for (int q = 0; q < r.hs; q++) out[i * r.hs + q] = near[i]; }
                row[k] = out;
            }
            if (++r.ystep >= r.vs) {
                r.ystep = 0; r.line0 = r.line1;
                if (++r.ypos < z.comp[k].y) r.line1 += z.comp[k].w2;
            }
        }
        uint8_t* o = im.px.data() + (size_t)j * im.w * 4;
        if (z.ncomp == 1) { for (int i = 0; i < im.w; i++, o += 4) { o[0] = o[1] = o[2] = row[0][i]; o[3] = 255; } }
        else if (is_rgb) { for (int i = 0; i < im.w; i++, o += 4) { o[0] = row[0][i]; o[1] = row[1][i]; o[2] = row[2][i]; o[3] = 255; } }
        else {
            auto fx = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
            static const int cr_r = fx(1.40200f), cr_g = -fx(0.71414f), cb_g = -fx(0.34414f), cb_b = fx(1.77200f);
            for (int i = 0; i < im.w; i++, o += 4) {
                const int y_fixed = (row[0][i] << 20) + (1 << 19);
                const int cr = row[2][i] - 128, cb = row[1][i] - 128;
                int rr = y_fixed + cr * cr_r;
                int gg = y_fixed + cr * cr_g + (int)((unsigned)(cb * cb_g) & 0xffff0000u);
                int bb = y_fixed + cb * cb_b;
                rr >>= 20; gg >>= 20; bb >>= 20;
                o[0] = JpegDec::clamp8(rr); o[1] = JpegDec::clamp8(gg); o[2] = JpegDec::clamp8(bb); o[3] = 255;
            }
        }
    }
    return im;
}

// what the image bytes are decides the decoder, as in stb_image (the glTF mimeType is not consulted there either)
Image decode_image(const uint8_t* d, size_t n) {
    if (n >= 2 && d[0] == 0xff && d[1] == 0xd8) return decode_jpeg(d, n);
    return decode_png(d, n);
}

// ---------------------------------------------------------------- textures as the reference stores them (Texture.h, USE_CHAR 1)
struct Tex { int w = 0, h = 0; std::vector<uint8_t> rgb; bool normal_transform = false; float sx = 1, sy = 1; };
inline uint8_t to_uchar(float v) { return (uint8_t)(int)v; }                      // float -> unsigned char, truncation
enum Conv { CONV_XYZ, CONV_SRGB };

// ImageHandler::apply_gamma(in, 1 / 2.2) (ImageProcessing.h:128-133 via FileLoading.h:361-363)
inline V3 gamma_srgb(V3 in) {
    const float gamma = (float)(1 / 2.2);
    const float lums = in.X * 0.2126f + in.Y * 0.7152f + in.Z * 0.0722f;
    if (lums == 0) return {0, 0, 0};
    const float post = (float)std::pow((double)lums, 1.0 / (double)gamma);
    return {in.X * post / lums, in.Y * post / lums, in.Z * post / lums};
}

// Texture::load (Texture.h:36-67,94-102): `position += components` BEFORE the read
Tex load_texture(const Image& im, Conv conv, bool with_scale, float scale_v) {
    Tex t; t.w = im.w; t.h = im.h; t.rgb.assign((size_t)im.w * im.h * 3, 0);
    t.normal_transform = with_scale;
    std::vector<uint8_t> img(im.px);
    img.resize(img.size() + 8, 0);                     // the read one texel past the end
    size_t position = 0;
    for (int y = 0; y < im.h; y++)
        for (int x = 0; x < im.w; x++) {
            position += (size_t)im.comp;
            const float r = ((float)img[position + 0]) / 255.0f, g = ((float)img[position + 1]) / 255.0f, b = ((float)img[position + 2]) / 255.0f;
            V3 v{r, g, b};
            if (conv == CONV_SRGB) v = gamma_srgb(v);
            uint8_t* o = &t.rgb[((size_t)y * im.w + x) * 3];
            if (with_scale) {                          // load_with_scale + commit_char with use_normal_transform
                const V3 in2 = sub(scale(v, 2.0f), V3{1, 1, 1});
                const V3 nv = normalize(mul(in2, V3{scale_v, scale_v, 1.0f}));
                o[0] = to_uchar(nv.X * 127 + 128); o[1] = to_uchar(nv.Y * 127 + 128); o[2] = to_uchar(nv.Z * 127 + 128);
            } else {
                o[0] = to_uchar(v.X * 255); o[1] = to_uchar(v.Y * 255); o[2] = to_uchar(v.Z * 255);
            }
        }
    return t;
}
// Texture::export_channel (Texture.h:214-226): value of one channel, re-quantised into byte 0 of a 3-byte texel
Tex export_channel(const Tex& src, int channel) {
    Tex t; t.w = src.w; t.h = src.h; t.sx = src.sx; t.sy = src.sy; t.rgb.assign(src.rgb.size(), 0);
    for (size_t i = 0; i < (size_t)src.w * src.h; i++) {
        const float v = src.rgb[3 * i + channel] / 255.0f;                        // fetch_char_XYZ (:147-155)
        t.rgb[3 * i] = to_uchar(v * 255);                                         // commit_char(float) (:133-136)
    }
    return t;
}

}  // namespace

struct srx_gltf {
    std::vector<srx_tri> tris;
    std::vector<srx_material> materials;
    std::vector<Tex> tex;
    std::vector<srx_texture> textures;
    std::vector<srx_sun> suns;
    srx_camera camera;
};

namespace {

struct Loader {
    JV root;
    std::vector<uint8_t> bin;
    srx_gltf* out;

    // A JSON number that is to be a byte offset / length / count: finite, integral, >= 0 and no larger than `limit`.
    // (A crafted file can hold -8 or 2^61 here; casting those to size_t and adding them wraps past a plain range check.)
    static size_t to_size(double v, size_t limit, const char* what) {
        if (!(v >= 0.0) || v != std::floor(v) || v > (double)limit) fail(std::string("glTF: ") + what + " is negative, fractional or beyond the file");
        return (size_t)v;
    }

    // ---- accessors (FileLoading.h:231-347)
    struct View { const uint8_t* p; size_t stride; size_t count; int comp_type; int ncomp; };
    View view(int accessor, int want_comp_type_a, int want_comp_type_b, int want_ncomp) {
        const JV& acc = root.get("accessors") ? root.get("accessors")->at((size_t)accessor) : JV();
        if (acc.t != JV::OBJ) fail("glTF: accessor index out of range");
        const int ct = acc.integer("componentType", 0);
        if (ct != want_comp_type_a && ct != want_comp_type_b) fail("glTF: accessor has an unsupported component type (floats for vertex data, u16/u32 for indices)");
        const std::string ty = acc.str("type");
        const int nc = ty == "SCALAR" ? 1 : ty == "VEC2" ? 2 : ty == "VEC3" ? 3 : ty == "VEC4" ? 4 : 0;
        if (nc != want_ncomp) fail("glTF: accessor has the wrong element type (" + ty + ")");
        const JV& bv = root.get("bufferViews") ? root.get("bufferViews")->at((size_t)acc.integer("bufferView", -1)) : JV();
        if (bv.t != JV::OBJ) fail("glTF: accessor without a bufferView");
        if (bv.integer("buffer", 0) != 0) fail("glTF: only the binary chunk of the .glb is supported as a buffer");
        const size_t csz = (ct == 5123) ? 2 : 4;
        const size_t elem = csz * nc;
        const size_t off = to_size(bv.number("byteOffset", 0), bin.size(), "bufferView byteOffset") +
                           to_size(acc.number("byteOffset", 0), bin.size(), "accessor byteOffset");      // each <= bin.size(): no wrap
        const size_t stride = bv.has("byteStride") ? to_size(bv.number("byteStride", 0), bin.size(), "byteStride") : elem;
        const size_t count = to_size(acc.number("count", 0), bin.size(), "accessor count");              // an element has at least a byte
        // off + stride * (count - 1) + elem <= bin.size(), written so that nothing can overflow
        if (count) {
            if (off > bin.size() || elem > bin.size() - off) fail("glTF: accessor reaches beyond the binary chunk");
            if (count > 1 && (stride == 0 ? false : (count - 1) > (bin.size() - off - elem) / stride)) fail("glTF: accessor reaches beyond the binary chunk");
        }
        return {bin.data() + off, stride, count, ct, nc};
    }
    std::vector<V3> vec3(int accessor) {
        View v = view(accessor, 5126, 5126, 3);
        std::vector<V3> r(v.count);
        for (size_t i = 0; i < v.count; i++) std::memcpy(&r[i], v.p + i * v.stride, 12);
        return r;
    }
    std::vector<float> vec2(int accessor) {
        View v = view(accessor, 5126, 5126, 2);
        std::vector<float> r(v.count * 2);
        for (size_t i = 0; i < v.count; i++) std::memcpy(&r[2 * i], v.p + i * v.stride, 8);
        return r;
    }
    std::vector<uint32_t> indices(int accessor) {
        View v = view(accessor, 5125, 5123, 1);
        std::vector<uint32_t> r(v.count);
        for (size_t i = 0; i < v.count; i++) {
            if (v.comp_type == 5125) std::memcpy(&r[i], v.p + i * v.stride, 4);
            else { uint16_t s; std::memcpy(&s, v.p + i * v.stride, 2); r[i] = s; }
        }
        return r;
    }

    // ---- textures (TextureLoader, FileLoading.h:17-165; fetch_transform :410-437)
    std::map<std::pair<int, std::pair<int, float>>, int> tex_cache;     // (gltf texture, converter, scale) -> base Tex index
    std::vector<Tex> base;
    int base_texture(int t_index, Conv conv, bool with_scale, float scale_v) {
        auto key = std::make_pair(t_index, std::make_pair((int)conv + (with_scale ? 2 : 0), with_scale ? scale_v : -9999.0f));
        auto it = tex_cache.find(key);
        if (it != tex_cache.end()) return it->second;
        const JV& td = root.get("textures") ? root.get("textures")->at((size_t)t_index) : JV();
        if (td.t != JV::OBJ) fail("glTF: texture index out of range");
        const JV& im = root.get("images") ? root.get("images")->at((size_t)td.integer("source", -1)) : JV();
        if (im.t != JV::OBJ) fail("glTF: texture without an image");
        if (!im.has("bufferView")) fail("glTF: image \"" + im.str("name") + "\" is not embedded in the .glb (uri images are not read)");
        const JV& bv = root.get("bufferViews") ? root.get("bufferViews")->at((size_t)im.integer("bufferView", -1)) : JV();
        if (bv.t != JV::OBJ) fail("glTF: image bufferView out of range");
        const size_t off = to_size(bv.number("byteOffset", 0), bin.size(), "image byteOffset");
        const size_t len = to_size(bv.number("byteLength", 0), bin.size(), "image byteLength");
        if (len > bin.size() - off) fail("glTF: image reaches beyond the binary chunk");
        const Image img = decode_image(bin.data() + off, len);
        base.push_back(load_texture(img, conv, with_scale, scale_v));
        return tex_cache[key] = (int)base.size() - 1;
    }
    static void transform_of(const JV* info, float& sx, float& sy) {
        sx = 1; sy = 1;
        if (!info) return;
        const JV* ext = info->get("extensions");
        const JV* tt = ext ? ext->get("KHR_texture_transform") : nullptr;
        if (!tt) return;
        const JV* sc = tt->get("scale");
        static const JV nul;
        sx = (float)(sc ? sc->at(0) : nul).as_double(); sy = (float)(sc ? sc->at(1) : nul).as_double();
    }
    int add_texture(Tex t) { out->tex.push_back(std::move(t)); return (int)out->tex.size() - 1; }

    std::vector<int> uv_map_index;     // Material::UV_map_index per material

    void materials() {
        const JV* ms = root.get("materials");
        for (size_t mi = 0; ms && mi < ms->size(); mi++) {
            const JV& md = ms->at(mi);
            static const JV empty;
            const JV& pbr = md.get("pbrMetallicRoughness") ? *md.get("pbrMetallicRoughness") : empty;
            srx_material m;
            std::memset(&m, 0, sizeof(m));
            m.color_tex = m.roughness_tex = m.metallic_tex = m.specular_tex = m.emissive_tex = m.normal_tex = m.transmission_tex = m.ior_tex = -1;
            const JV* bc = pbr.get("baseColorFactor");
            for (int k = 0; k < 3; k++) m.color[k] = bc ? (float)bc->at(k).as_double() : 1.0f;                  // :470
            const double metallic_factor = pbr.number("metallicFactor", 1.0), roughness_factor = pbr.number("roughnessFactor", 1.0);
            m.metallic = (float)metallic_factor; m.roughness = (float)roughness_factor;                          // :471-472
            const JV* em = md.get("emissiveFactor");
            for (int k = 0; k < 3; k++) m.emissive[k] = (em ? (float)em->at(k).as_double() : 0.0f) * 25;         // :473
            int uv_index = 0;
            const JV* bct = pbr.get("baseColorTexture");
            if (bct && bct->integer("index", -1) != -1) {                                                        // :477-487
                uv_index = bct->integer("texCoord", 0);
                Tex t = base[base_texture(bct->integer("index", -1), CONV_SRGB, false, 0)];
                transform_of(bct, t.sx, t.sy);
                m.color_tex = add_texture(std::move(t));
            }
            const JV* mrt = pbr.get("metallicRoughnessTexture");
            if (mrt && mrt->integer("index", -1) != -1) {                                                        // :488-518
                uv_index = bct ? bct->integer("texCoord", 0) : 0;
                Tex t = base[base_texture(mrt->integer("index", -1), CONV_XYZ, false, 0)];
                transform_of(mrt, t.sx, t.sy);
                Tex metallic = export_channel(t, 2), roughness = export_channel(t, 1);
                if (metallic_factor != 0) {
                    if (metallic_factor != 1)
                        for (size_t i = 0; i < (size_t)metallic.w * metallic.h; i++) metallic.rgb[i] = (uint8_t)(int)(metallic.rgb[i] * metallic_factor);
                    m.metallic_tex = add_texture(std::move(metallic));
                }
                if (roughness_factor != 0) m.roughness_tex = add_texture(std::move(roughness));
            }
            const JV* nt = md.get("normalTexture");
            if (nt && nt->integer("index", -1) != -1) {                                                          // :519-534
                uv_index = nt->integer("texCoord", 0);
                const float s = (float)nt->number("scale", 1.0);
                Tex t = base[base_texture(nt->integer("index", -1), CONV_XYZ, true, s)];
                transform_of(nt, t.sx, t.sy);
                m.normal_tex = add_texture(std::move(t));
                m.use_normals = 1;
            }
            const JV* ext = md.get("extensions");
            if (ext && ext->has("KHR_materials_specular")) {                                                     // :535-552
                const JV* f = ext->get("KHR_materials_specular")->get("specularColorFactor");
                m.specular = (f && f->t != JV::NUL) ? (float)f->at(0).as_double() : 0.0f;
            } else m.specular = 1;
            if (ext && ext->has("KHR_materials_ior")) {                                                          // :553-563
                const JV* f = ext->get("KHR_materials_ior")->get("ior");
                m.ior = (float)(f ? f->as_double() : 0.0);
            } else m.ior = 1.5;
            if (ext && ext->has("KHR_materials_transmission")) {                                                 // :564-574
                const JV* f = ext->get("KHR_materials_transmission")->get("transmissionFactor");
                m.transmission = (float)(f ? f->as_double() : 0.0);
            } else m.transmission = 0;
            out->materials.push_back(m);
            uv_map_index.push_back(uv_index);
        }
        srx_material d;                                                                                          // :579-587
        std::memset(&d, 0, sizeof(d));
        d.color_tex = d.roughness_tex = d.metallic_tex = d.specular_tex = d.emissive_tex = d.normal_tex = d.transmission_tex = d.ior_tex = -1;
        d.roughness = 0.5f; d.ior = 1.5f; d.color[0] = d.color[1] = d.color[2] = 1; d.metallic = 0;
        out->materials.push_back(d);
        uv_map_index.push_back(0);
    }

    std::vector<std::vector<Tri>> meshes;
    void load_meshes() {                                                                                         // :588-674
        const int default_index = (int)out->materials.size() - 1;
        const JV* ms = root.get("meshes");
        for (size_t mi = 0; ms && mi < ms->size(); mi++) {
            std::vector<Tri> tris;
            const JV* prims = ms->at(mi).get("primitives");
            for (size_t pi = 0; prims && pi < prims->size(); pi++) {
                const JV& pr = prims->at(pi);
                static const JV empty;
                const JV& attr = pr.get("attributes") ? *pr.get("attributes") : empty;
                int mat = pr.integer("material", -1);
                if (mat <= -1) mat = default_index;
                if (mat >= (int)out->materials.size() - 1 && mat != default_index) fail("glTF: primitive material out of range");
                if (!attr.has("POSITION")) fail("glTF: primitive without POSITION");
                if (pr.has("mode") && pr.integer("mode", 4) != 4) fail("glTF: only triangle lists (mode 4) are read");
                const int pos_acc = attr.integer("POSITION", 0);
                const std::string tc = "TEXCOORD_" + std::to_string(uv_map_index[(size_t)mat]);
                const int tex_acc = attr.integer(tc.c_str(), 0), nrm_acc = attr.integer("NORMAL", 0);
                const std::vector<V3> P = vec3(pos_acc);
                const bool uv_mapped = tex_acc != 0, smooth = nrm_acc != 0;                                       // :614,620
                std::vector<float> UV; std::vector<V3> N;
                if (uv_mapped) UV = vec2(tex_acc);
                if (smooth) N = vec3(nrm_acc);
                if (!pr.has("indices")) fail("glTF: primitive without indices");
                const std::vector<uint32_t> idx = indices(pr.integer("indices", -1));
                if (idx.size() % 3) fail("glTF: index count is not a multiple of 3");
                for (size_t i = 0; i < idx.size() / 3; i++) {
                    const uint32_t l[3] = {idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]};
                    float uv[6] = {0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < 3; k++) {
                        if (l[k] >= P.size() || (uv_mapped && l[k] * 2 + 1 >= UV.size()) || (smooth && l[k] >= N.size())) fail("glTF: vertex index out of range");
                        if (uv_mapped) { uv[2 * k] = 0.0f - (-1.0f * UV[2 * l[k]]); uv[2 * k + 1] = 1.0f - (1.0f * UV[2 * l[k] + 1]); }   // :658-660
                    }
                    Tri t = make_tri(swz(P[l[0]]), swz(P[l[1]]), swz(P[l[2]]), uv, mat);
                    if (smooth) { t.n1 = swz(N[l[0]]); t.n2 = swz(N[l[1]]); t.n3 = swz(N[l[2]]); }
                    tris.push_back(t);
                }
            }
            meshes.push_back(std::move(tris));
        }
    }

    struct Cam { V3 position; Q4 rotation{0, 0, 0, 1}; float aspect = 1, distance = 1; };
    std::vector<Cam> cameras;
    std::vector<Object> objects;
    std::vector<srx_sun> lights;
    std::map<int, std::pair<int, int>> node_lookup;

    void load_nodes() {
        const JV* cs = root.get("cameras");
        for (size_t i = 0; cs && i < cs->size(); i++) {                                                          // :677-685
            if (cs->at(i).str("type") != "perspective") fail("glTF: only perspective cameras are read");
            static const JV empty;
            const JV& p = cs->at(i).get("perspective") ? *cs->at(i).get("perspective") : empty;
            Cam c;
            c.aspect = (float)p.number("aspectRatio", 0.0);
            c.distance = (float)(0.5 / std::tan(0.5 * p.number("yfov", 0.0)));
            cameras.push_back(c);
        }
        const JV* ext = root.get("extensions");
        const JV* lp = ext ? ext->get("KHR_lights_punctual") : nullptr;
        const JV* light_defs = lp ? lp->get("lights") : nullptr;
        const JV* ns = root.get("nodes");
        for (size_t ni = 0; ns && ni < ns->size(); ni++) {                                                       // :686-742
            const JV& nd = ns->at(ni);
            const JV* sc = nd.get("scale"); const JV* tr = nd.get("translation"); const JV* ro = nd.get("rotation");
            const bool has_s = sc && sc->size() > 0, has_t = tr && tr->size() > 0, has_r = ro && ro->size() > 0;
            const int cam = nd.integer("camera", -1), mesh = nd.integer("mesh", -1);
            const JV* next = nd.get("extensions");
            const JV* nl = next ? next->get("KHR_lights_punctual") : nullptr;
            const int light = nl ? nl->integer("light", -1) : -1;
            if (cam > -1) {
                if ((size_t)cam >= cameras.size()) fail("glTF: node camera out of range");
                if (has_t) cameras[(size_t)cam].position = add(cameras[(size_t)cam].position, swz_pos(*tr));
                cameras[(size_t)cam].rotation = has_r ? swz_rot(*ro) : Q4{0, 0, 0, 0};                           // Quat() : W(0)
                node_lookup[(int)ni] = {0, cam};
                continue;
            }
            if (mesh > -1) {
                if ((size_t)mesh >= meshes.size()) fail("glTF: node mesh out of range");
                Object o;
                if (has_s) o.scale = swz_scale(*sc);
                Transform r; r.has_rot = true; r.rot = has_r ? swz_rot(*ro) : Q4{0, 0, 0, 1};
                Transform m; m.has_move = true; m.move = has_t ? swz_pos(*tr) : V3{0, 0, 0};
                o.transformations.push_back(r); o.transformations.push_back(m);                                  // :711-712
                o.mesh = mesh;
                objects.push_back(o);
                node_lookup[(int)ni] = {1, (int)objects.size() - 1};
                continue;
            }
            if (light > -1) {
                const JV& ld = light_defs ? light_defs->at((size_t)light) : JV();
                if (ld.t != JV::OBJ) fail("glTF: node light out of range");
                const Q4 rot = has_r ? swz_rot(*ro) : Q4{0, 0, 0, 1};
                if (ld.str("type") == "directional") {                                                           // :724-732
                    const V3 dir = apply_rotation(V3{0, -1, 0}, rot);
                    const JV* col = ld.get("color");
                    const float intensity = (float)ld.number("intensity", 1.0);
                    srx_sun s;
                    const float d3[3] = {dir.X, dir.Y, dir.Z};
                    for (int k = 0; k < 3; k++) { s.direction[k] = d3[k]; s.emission[k] = (col ? (float)col->at(k).as_double() : 1.0f) * intensity / 50000; }
                    lights.push_back(s);
                    node_lookup[(int)ni] = {2, (int)lights.size() - 1};
                } else fail("glTF: light type \"" + ld.str("type") + "\" is not supported (the reference only builds SunLight)");
                continue;
            }
        }
        for (auto& kv : node_lookup) {                                                                           // :743-755 (mesh nodes only)
            if (kv.second.first != 1) continue;
            const JV* ch = ns->at((size_t)kv.first).get("children");
            for (size_t c = 0; ch && c < ch->size(); c++) {
                auto it = node_lookup.find((int)ch->at(c).as_double());
                if (it == node_lookup.end() || it->second.first != 1) continue;
                // a node hierarchy is a forest (glTF 2.0, 3.5.2): a second parent or a cycle would make the walks below endless
                if (objects[(size_t)it->second.second].parent >= 0) fail("glTF: a mesh node has two parents");
                for (int up = kv.second.second; up >= 0; up = objects[(size_t)up].parent)
                    if (up == it->second.second) fail("glTF: the node hierarchy has a cycle");
                objects[(size_t)kv.second.second].children.push_back(it->second.second);
                objects[(size_t)it->second.second].parent = kv.second.second;
            }
        }
    }

    V3 compile_scale(int oi, V3 s) {                                                                             // Primitives.h:462-471
        s = mul(s, objects[(size_t)oi].scale);
        if (objects[(size_t)oi].parent >= 0) s = mul(s, compile_scale(objects[(size_t)oi].parent, s));
        return s;
    }
    void compile_transforms(int oi, std::vector<Transform>& in) {                                                // :472-481
        for (auto& t : objects[(size_t)oi].transformations) in.push_back(t);
        if (objects[(size_t)oi].parent >= 0) compile_transforms(objects[(size_t)oi].parent, in);
    }
    void fetch_data(int oi, int depth) {                                                                         // :423-434, Packers::transformT :347-373
        if (depth > 256) fail("glTF: node hierarchy too deep (a cycle?)");
        const Object& o = objects[(size_t)oi];
        const V3 scale_final = compile_scale(oi, V3{1, 1, 1});
        std::vector<Transform> all;
        compile_transforms(oi, all);
        V3 position{0, 0, 0}; Q4 rotation{0, 0, 0, 1};
        for (auto& t : all) stack(t, position, rotation);                                                        // Transformation::collpase
        for (const Tri& t : meshes[(size_t)o.mesh]) {
            V3 np[3] = {sub(t.p1, o.origin), sub(t.p2, o.origin), sub(t.p3, o.origin)};
            for (auto& p : np) { p = mul(p, scale_final); p = apply_rotation(p, rotation); p = add(p, position); p = add(p, o.origin); }
            const V3 nn[3] = {apply_rotation(t.n1, rotation), apply_rotation(t.n2, rotation), apply_rotation(t.n3, rotation)};
            srx_tri s;
            for (int k = 0; k < 3; k++) {
                s.p[3 * k] = np[k].X; s.p[3 * k + 1] = np[k].Y; s.p[3 * k + 2] = np[k].Z;
                s.n[3 * k] = nn[k].X; s.n[3 * k + 1] = nn[k].Y; s.n[3 * k + 2] = nn[k].Z;
            }
            std::memcpy(s.uv, t.uv, sizeof(s.uv));
            s.material = t.material; s.flags = 1u;       // the rotated normals of the ORIGINAL triangle are kept (transformT)
            out->tris.push_back(s);
        }
        for (int c : o.children) fetch_data(c, depth + 1);
    }

    void run() {
        materials();
        load_meshes();
        load_nodes();
        // scene registration in node order (:759-777; an index that is no camera / mesh / light node registers camera 0)
        int camera = -1, last = -1;
        for (auto& kv : node_lookup) last = std::max(last, kv.first);
        std::vector<int> roots;
        for (int i = 0; i <= last; i++) {
            auto it = node_lookup.find(i);
            const int type = it == node_lookup.end() ? 0 : it->second.first, index = it == node_lookup.end() ? 0 : it->second.second;
            if (type == 0 && camera < 0 && !cameras.empty()) camera = index;
            if (type == 1 && objects[(size_t)index].parent < 0) roots.push_back(index);
            if (type == 2) out->suns.push_back(lights[(size_t)index]);
        }
        if (camera < 0) fail("glTF: the scene has no camera node (the reference takes scene->cameras[0])");
        for (int r : roots) fetch_data(r, 0);                                                                    // Scene::package, Scene.h:104-106
        const Cam& c = cameras[(size_t)camera];
        std::memset(&out->camera, 0, sizeof(out->camera));
        out->camera.position[0] = c.position.X; out->camera.position[1] = c.position.Y; out->camera.position[2] = c.position.Z;
        out->camera.rotation[0] = c.rotation.X; out->camera.rotation[1] = c.rotation.Y; out->camera.rotation[2] = c.rotation.Z; out->camera.rotation[3] = c.rotation.W;
        out->camera.focal_position[1] = c.distance;                                                              // Camera(XYZ(), lens, XYZ(0, distance, 0)), :683
        out->camera.lens_width = c.aspect; out->camera.lens_height = 1;                                          // RectLens(aspect_ratio, 1), :681
        for (auto& t : out->tex) {
            srx_texture st;
            st.rgb8 = t.rgb.data(); st.width = t.w; st.height = t.h; st.scale_x = t.sx; st.scale_y = t.sy; st.normal_transform = t.normal_transform ? 1 : 0;
            out->textures.push_back(st);
        }
    }
};

}  // namespace

extern "C" int srx_gltf_load(const char* path, srx_gltf** out) {
    if (!path || !out) { srx_set_error("srx_gltf_load: null argument"); return 1; }
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) { srx_set_error(std::string("srx_gltf_load: cannot open ") + path); return 1; }
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    for (size_t n; (n = std::fread(buf, 1, sizeof(buf), f)) > 0;) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    std::unique_ptr<srx_gltf> g(new srx_gltf());
    try {
        auto u32 = [&](size_t at) { uint32_t v; std::memcpy(&v, file.data() + at, 4); return v; };
        if (file.size() < 20 || u32(0) != 0x46546C67u) fail("not a binary glTF (.glb) file (the reference loads .glb only, FileLoading.h:208)");
        if (u32(4) != 2) fail("glTF container version is not 2");
        Loader L;
        L.out = g.get();
        bool have_json = false;
        for (size_t at = 12; at + 8 <= file.size();) {
            const uint32_t len = u32(at), type = u32(at + 4);
            if (at + 8 + (size_t)len > file.size()) fail("glTF chunk reaches beyond the file");
            if (type == 0x4E4F534Au) { JParser P{(const char*)file.data() + at + 8, (const char*)file.data() + at + 8 + len}; L.root = P.parse(); have_json = true; }
            else if (type == 0x004E4942u && L.bin.empty()) L.bin.assign(file.begin() + at + 8, file.begin() + at + 8 + len);
            at += 8 + (size_t)len;
        }
        if (!have_json || L.root.t != JV::OBJ) fail("glTF file has no JSON chunk");
        L.run();
    } catch (const Fail& e) {
        srx_set_error(std::string("srx_gltf_load: ") + path + ": " + e.msg);
        return 1;
    } catch (const std::exception& e) {
        srx_set_error(std::string("srx_gltf_load: ") + path + ": " + e.what());
        return 1;
    }
    *out = g.release();
    return 0;
}

extern "C" void srx_gltf_destroy(srx_gltf* g) { delete g; }

extern "C" int srx_image_decode(const uint8_t* data, size_t n, uint32_t* width, uint32_t* height, uint8_t* rgba, size_t capacity) {
    if (!data || !width || !height) { srx_set_error("srx_image_decode: null argument"); return 1; }
    try {
        const Image im = decode_image(data, n);
        *width = (uint32_t)im.w; *height = (uint32_t)im.h;
        if (rgba) {
            if (capacity < im.px.size()) { srx_set_error("srx_image_decode: output buffer smaller than width * height * 4"); return 1; }
            std::memcpy(rgba, im.px.data(), im.px.size());
        }
        return 0;
    } catch (const Fail& e) {
        srx_set_error(std::string("srx_image_decode: ") + e.msg);
        return 1;
    } catch (const std::exception& e) {
        srx_set_error(std::string("srx_image_decode: ") + e.what());
        return 1;
    }
}

extern "C" int srx_gltf_get(const srx_gltf* g, const srx_tri** tris, uint32_t* n_tris, const srx_material** materials, uint32_t* n_materials,
                            const srx_texture** textures, uint32_t* n_textures, const srx_sun** suns, uint32_t* n_suns, srx_camera* camera) {
    if (!g) { srx_set_error("srx_gltf_get: null handle"); return 1; }
    if (tris) *tris = g->tris.data();
    if (n_tris) *n_tris = (uint32_t)g->tris.size();
    if (materials) *materials = g->materials.data();
    if (n_materials) *n_materials = (uint32_t)g->materials.size();
    if (textures) *textures = g->textures.data();
    if (n_textures) *n_textures = (uint32_t)g->textures.size();
    if (suns) *suns = g->suns.data();
    if (n_suns) *n_suns = (uint32_t)g->suns.size();
    if (camera) *camera = g->camera;
    return 0;
}
