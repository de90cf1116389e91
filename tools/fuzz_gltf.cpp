// tools/fuzz_gltf.cpp — mutation fuzz of the glTF loader and the packager under host ASan + UBSan (tools/fuzz_gltf.sh).
#include "srx.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <random>
// mutate a good .glb many ways, load each under ASan/UBSan; also run the packager on whatever loads
static std::vector<unsigned char> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<unsigned char> v; if (!f) return v; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v; }
static void spit(const char* p, const std::vector<unsigned char>& v) { FILE* f = fopen(p, "wb"); fwrite(v.data(), 1, v.size(), f); fclose(f); }
int main(int argc, char** argv) {
    int iters = argc > 2 ? atoi(argv[2]) : 1000;
    std::vector<unsigned char> good = slurp(argv[1]);
    if (good.empty()) { printf("no input\n"); return 2; }
    std::mt19937 rng(12345);
    int ok = 0, bad = 0, packed = 0;
    if (good.size() < 4 || memcmp(good.data(), "glTF", 4)) {
        // an image file (PNG / JPEG): the decoders on their own (srx_image_decode), same mutations minus the JSON ones
        std::vector<unsigned char> px;
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> v = good;
            const int kind = it == 0 ? -1 : it % 4;
            if (kind == 0) v.resize(rng() % v.size());
            else if (kind == 1) { int n = 1 + rng() % 8; for (int k = 0; k < n; k++) v[rng() % v.size()] ^= (unsigned char)(1u << (rng() % 8)); }
            else if (kind == 2) { size_t p = rng() % v.size(); size_t n = rng() % 32; for (size_t k = 0; k < n && p + k < v.size(); k++) v[p + k] = (unsigned char)rng(); }
            else if (kind == 3) { v[rng() % v.size()] = (unsigned char)rng(); }
            uint32_t w = 0, h = 0;
            if (srx_image_decode(v.data(), v.size(), &w, &h, nullptr, 0) == 0 && (uint64_t)w * h <= (1u << 22)) {
                px.resize((size_t)w * h * 4);
                if (srx_image_decode(v.data(), v.size(), &w, &h, px.data(), px.size()) == 0) ok++; else bad++;
            } else bad++;
        }
        printf("decoded %d refused %d\n", ok, bad);
        return 0;
    }
    for (int it = 0; it < iters; it++) {
        std::vector<unsigned char> v = good;
        int kind = it % 6;
        if (it == 0) kind = -1;
        if (kind == 0) v.resize(rng() % v.size());
        else if (kind == 1) { int n = 1 + rng() % 8; for (int k = 0; k < n; k++) v[rng() % v.size()] ^= (unsigned char)(1u << (rng() % 8)); }
        else if (kind == 2) { // mutate inside the JSON chunk: replace a digit
            uint32_t jl; memcpy(&jl, &v[12], 4); if (jl + 20 <= v.size()) { for (int k = 0; k < 4; k++) { size_t p = 20 + rng() % jl; if (v[p] >= '0' && v[p] <= '9') v[p] = (unsigned char)('0' + rng() % 10); } } }
        else if (kind == 3) { size_t p = rng() % v.size(); size_t n = rng() % 64; for (size_t k = 0; k < n && p + k < v.size(); k++) v[p + k] = (unsigned char)rng(); }
        else if (kind == 4) { uint32_t jl; memcpy(&jl, &v[12], 4); if (jl + 20 <= v.size()) { size_t p = 20 + rng() % jl; v[p] = (unsigned char)"{}[],:\"-e."[rng() % 10]; } }
        else if (kind == 5) { // big numbers in JSON
            uint32_t jl; memcpy(&jl, &v[12], 4); if (jl + 20 <= v.size()) { for (int k = 0; k < 2; k++) { size_t p = 20 + rng() % jl; if (v[p] >= '0' && v[p] <= '9') v[p] = '9'; } } }
        spit("/tmp/srx_fuzz_cur.glb", v);
        srx_gltf* g = nullptr;
        if (srx_gltf_load("/tmp/srx_fuzz_cur.glb", &g) == 0) {
            ok++;
            const srx_tri* tris; uint32_t nt, nm, ntx, ns; const srx_material* m; const srx_texture* tx; const srx_sun* su; srx_camera cam;
            srx_gltf_get(g, &tris, &nt, &m, &nm, &tx, &ntx, &su, &ns, &cam);
            if (nt > 0 && nt < 200000) { srx_package* pk = nullptr; if (srx_package_create(tris, nt, &pk) == 0) { packed++; srx_package_destroy(pk); } }
            srx_gltf_destroy(g);
        } else bad++;
    }
    printf("loaded %d refused %d packaged %d\n", ok, bad, packed);
    return 0;
}
