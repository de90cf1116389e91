// oracle/sro_scene.h — TEST INFRASTRUCTURE ONLY (the parity oracle).
//
// CPU restatement of the reference scene packager: authoring triangles ->
// binary SAH-ish BVH -> 8-wide collapse -> 8-triangle packets. The order of
// tri_data (and therefore every tri_index the casts return) is a by-product of
// this code, so it follows the reference statement by statement, including the
// std::sort calls whose tie order belongs to libstdc++.
// PARITY UNPINNED (no upstream tests; see sro_math.h header and DESIGN.md).
#pragma once
#include "sro_math.h"
#include <queue>
#include <map>
#include <stdexcept>

namespace sro {

// ---- reference record layouts (a2, a5, a11), byte-exact -----
#pragma pack(push, 1)
struct PackagedTri {           // Primitives.h:62-81, 196 B
    XYZ p1, p1p3, p1p2;
    XY UV_basis, U_delta, V_delta;
    uint32_t material;
    XYZ normal, origin_offset;
    XYZ n1, n2, n3, t1, t2, t3, b1, b2, b3;
};
struct BVH_AVX {               // BVH.h:874-881, 288 B
    float maxX[8], maxY[8], maxZ[8];
    float minX[8], minY[8], minZ[8];
    uint32_t leaf_size[8];
    uint32_t indexes[8];
    int32_t parent_index, self_index;
    uint8_t pad[24];
};
struct PTri_AVX {              // Primitives.h:127-134, 448 B
    float p1X[8], p1Y[8], p1Z[8];
    float p1p2X[8], p1p2Y[8], p1p2Z[8];
    float p1p3X[8], p1p3Y[8], p1p3Z[8];
    XYZ origin_offset[8];
    uint32_t materials[8];
    int32_t index;
    int8_t size;
    uint8_t pad[27];
};
#pragma pack(pop)
static_assert(sizeof(PackagedTri) == 196, "PackagedTri");
static_assert(sizeof(BVH_AVX) == 288, "BVH_AVX");
static_assert(sizeof(PTri_AVX) == 448, "PTri_AVX");

// ---- authoring triangle (Primitives.h:224-258), world space -------------
struct Tri {
    XYZ p1, p2, p3;
    XY UV_1, UV_2, UV_3;
    XYZ n1, n2, n3;
    XYZ midpoint, AABB_max, AABB_min;
    uint32_t material;
    Tri(XYZ _p1, XYZ _p2, XYZ _p3, XY u1, XY u2, XY u3, short _material)
        : p1(_p1), p2(_p2), p3(_p3), material((uint32_t)(int)_material) {
        midpoint = (p1 + p2 + p3) / 3;
        AABB_max = XYZ::max(p1, XYZ::max(p2, p3));
        AABB_min = XYZ::min(p1, XYZ::min(p2, p3));
        UV_1 = u1; UV_2 = u2; UV_3 = u3;
        XYZ p1p3 = _p3 - p1;
        XYZ p1p2 = _p2 - p1;
        XYZ normal = XYZ::normalize(XYZ::cross(p1p3, p1p2));
        n1 = normal; n2 = normal; n3 = normal;
    }
    // Primitives.h:82-124 + 301-312
    PackagedTri pack() const {
        PackagedTri PT;
        std::memset((void*)&PT, 0, sizeof(PT));
        PT.material = (uint32_t)(int)(short)material;
        PT.p1 = p1;
        PT.p1p3 = p3 - p1;
        PT.p1p2 = p2 - p1;
        PT.U_delta = UV_2 - UV_1;
        PT.V_delta = UV_3 - UV_1;
        if (PT.U_delta == XY(0)) PT.U_delta = XY(1, 0);
        if (PT.V_delta == XY(0)) PT.V_delta = XY(0, 1);
        PT.UV_basis = UV_1;
        PT.normal = XYZ::normalize(XYZ::cross(PT.p1p3, PT.p1p2));
        PT.origin_offset = XYZ::dot((p1 + p2 + p3) / 3, PT.normal) * PT.normal;
        PT.n1 = n1; PT.n2 = n2; PT.n3 = n3;
        // calculate_tangents :107-111
        PT.t1 = XYZ::normalize(PT.p1p3 * PT.U_delta.X - PT.p1p2 * PT.V_delta.X);
        PT.t2 = PT.t1; PT.t3 = PT.t1;
        // calculate_bitangents :112-116
        PT.b1 = XYZ::normalize(PT.p1p2 * PT.V_delta.Y - PT.p1p3 * PT.U_delta.Y);
        PT.b2 = PT.b1; PT.b3 = PT.b1;
        // reorthogonalize :117-124 (quirk Q7: n1 in all three dots)
        PT.t1 = XYZ::normalize(PT.t1 - PT.n1 * XYZ::dot(PT.t1, PT.n1));
        PT.t2 = XYZ::normalize(PT.t2 - PT.n2 * XYZ::dot(PT.t2, PT.n1));
        PT.t3 = XYZ::normalize(PT.t3 - PT.n3 * XYZ::dot(PT.t3, PT.n1));
        PT.b1 = XYZ::cross(PT.n1, PT.t1);
        PT.b2 = XYZ::cross(PT.n2, PT.t2);
        PT.b3 = XYZ::cross(PT.n3, PT.t3);
        return PT;
    }
};

// VecLib.cpp:428-433
static inline float surface_area(const XYZ& max, const XYZ& min) {
    float l_x = max.X - min.X, l_y = max.Y - min.Y, l_z = max.Z - min.Z;
    return (l_x * l_y + l_x * l_z + l_y * l_z) * 2;
}

// ---- binary builder (BVH.h:17-696) --------------------------------------
struct BVH {
    XYZ max = XYZ(0, 0, 0), min = XYZ(0, 0, 0);
    std::vector<const Tri*> elements;
    BVH* c1 = nullptr;
    BVH* c2 = nullptr;
    int _count = -1;
    BVH() {}
    explicit BVH(const std::vector<const Tri*>& T_vec) {                     // :27-34 (quirk Q1)
        elements = T_vec;
        for (const Tri* T : T_vec) {
            max = XYZ::max(max, T->AABB_max);
            min = XYZ::min(min, T->AABB_min);
        }
    }
    ~BVH() { delete c1; delete c2; }
    int count() {                                                            // :217-229 (starts at -1)
        if (_count == -1) count(_count);
        return _count;
    }
    void count(int& s) const {
        s += (int)elements.size();
        if (c1 != nullptr) { c1->count(s); c2->count(s); }
    }
    int size() const { int s = 1; if (c1) { s += c1->size() + c2->size(); } return s; }

    struct Stats {                                                           // :259-269
        XYZ min = XYZ(99999999999.f), max = XYZ();
        int count = 0;
        float SA() const { return surface_area(max, min); }
    };
    struct Split {                                                           // :270-277
        Stats p, n;
        float score = 0;
        XYZ placement;
        int facing;
        Split(XYZ place, int f) : placement(place), facing(f) {}
    };
    struct relevant_value {                                                  // :331-345
        XYZ midpoint, max, min, right_max, right_min;
        const Tri* parent;
        explicit relevant_value(const Tri* src)
            : midpoint(src->midpoint), max(src->AABB_max), min(src->AABB_min), parent(src) {}
    };
    static float evaluate_split(Split& split, float SA_parent) {             // :676-694
        float traversal_cost = 1;
        float intersect_cost = 32;
        float Sa = split.p.SA() / SA_parent;
        float Sb = split.n.SA() / SA_parent;
        int count_a = split.p.count;
        int count_b = split.n.count;
        count_a = (int)std::ceil(count_a / (float)8);                        // PENALIZE_UNFILLED_LEAFS
        count_b = (int)std::ceil(count_b / (float)8);
        intersect_cost = 2;
        float Ha = Sa * count_a * intersect_cost;
        float Hb = Sb * count_b * intersect_cost;
        float final_h = traversal_cost + Ha + Hb;
        split.score = final_h;
        return final_h;
    }
    static void get_stats(const std::vector<const Tri*>& geo, Split& split) { // :659-675
        split.p.count = 0; split.n.count = 0;
        split.p.max = XYZ(-99999999.f); split.n.max = XYZ(-99999999.f);
        split.p.min = XYZ(9999999.f);   split.n.min = XYZ(9999999.f);
        int axis = split.facing;
        for (const Tri* T : geo) {
            if (T->midpoint[axis] > split.placement[axis]) {
                split.p.count++;
                split.p.max = XYZ::max(T->AABB_max, split.p.max);
                split.p.min = XYZ::min(T->AABB_min, split.p.min);
            }
            if (T->midpoint[axis] <= split.placement[axis]) {
                split.n.count++;
                split.n.max = XYZ::max(T->AABB_max, split.n.max);
                split.n.min = XYZ::min(T->AABB_min, split.n.min);
            }
        }
    }
    static Split probe(const std::vector<const Tri*>& geo) {                 // :581-653
        std::vector<relevant_value> sorted;
        sorted.reserve(geo.size());
        for (const Tri* T : geo) sorted.push_back(relevant_value(T));
        Split operator_split(XYZ(), 0);
        Split best = operator_split;
        best.score = 999999999999999999999999.0f;
        int geo_count = (int)geo.size();
        for (int i = 0; i < 3; i++) {
            operator_split.facing = i;
            std::sort(sorted.begin(), sorted.end(),
                      [i](const relevant_value& v1, const relevant_value& v2) {
                          return v1.midpoint[i] < v2.midpoint[i];
                      });
            XYZ left_min = sorted.front().min;
            XYZ left_max = sorted.front().max;
            XYZ right_min = sorted.back().min;
            XYZ right_max = sorted.back().max;
            int count = 0;
            for (int j = (int)sorted.size() - 1; j >= 0; j--) {
                relevant_value& v = sorted[j];
                right_min = XYZ::min(right_min, v.min);
                right_max = XYZ::max(right_max, v.max);
                v.right_min = right_min;
                v.right_max = right_max;
            }
            operator_split.n.min = XYZ();
            operator_split.n.max = XYZ();
            operator_split.n.count = 0;
            operator_split.p.min = right_min;
            operator_split.p.max = right_max;
            operator_split.p.count = geo_count;
            evaluate_split(operator_split, 1);
            if (operator_split.score < best.score) best = operator_split;
            for (size_t j = 0; j < sorted.size(); j++) {
                const relevant_value& v = sorted[j];
                left_min = XYZ::min(left_min, v.min);
                left_max = XYZ::max(left_max, v.max);
                count++;
                operator_split.n.min = left_min;
                operator_split.n.max = left_max;
                operator_split.n.count = count;
                operator_split.p.min = v.right_min;
                operator_split.p.max = v.right_max;
                operator_split.p.count = geo_count - count;
                operator_split.placement = v.midpoint;
                evaluate_split(operator_split, 1);
                if (operator_split.score < best.score) best = operator_split;
            }
        }
        return best;
    }
    // :98-180
    int construct(const std::vector<const Tri*>& initial_geo) {
        struct WorkPacket { BVH* target; std::vector<const Tri*>* contents; };
        std::queue<WorkPacket> packets;
        if (initial_geo.empty()) throw std::runtime_error("BVH creation attempted with zero geometry");
        packets.push({this, new std::vector<const Tri*>(initial_geo)});
        int i = 0;
        while (!packets.empty()) {
            i++;
            WorkPacket packet = packets.front();
            std::vector<const Tri*>* geo = packet.contents;
            BVH* target = packet.target;
            // getBounds :308-315
            XYZ bmax(-99999999999999.f), bmin(99999999999999.f);
            for (const Tri* T : *geo) {
                bmax = XYZ::max(bmax, T->AABB_max); bmin = XYZ::min(bmin, T->AABB_max);
                bmax = XYZ::max(bmax, T->AABB_min); bmin = XYZ::min(bmin, T->AABB_min);
            }
            target->max = bmax;
            target->min = bmin;
            XYZ extent = target->max - target->min;
            int axis = 0;
            if (extent.Y > extent.X) axis = 1;
            if (extent.Z > extent[axis]) axis = 2;
            float splitPos = target->min[axis] + extent[axis] * 0.5f;
            Split split(XYZ(splitPos), axis);
            get_stats(*geo, split);
            evaluate_split(split, 1);
            Split probe_split = probe(*geo);
            get_stats(*geo, probe_split);
            evaluate_split(probe_split, 1);
            if (probe_split.score < split.score) split = probe_split;
            // bin :316-330
            auto* p_geo = new std::vector<const Tri*>();
            auto* n_geo = new std::vector<const Tri*>();
            for (const Tri* T : *geo) {
                if (T->midpoint[split.facing] > split.placement[split.facing]) p_geo->push_back(T);
                else n_geo->push_back(T);
            }
            if (((size_t)std::abs((int)(p_geo->size() - n_geo->size()))) == geo->size()) {  // :145-149
                target->elements = *geo;
                packets.pop();
                delete p_geo; delete n_geo; delete geo;
                continue;
            }
            if (p_geo->size() <= 8) { target->c1 = new BVH(*p_geo); delete p_geo; }
            else { target->c1 = new BVH(); packets.push({target->c1, p_geo}); }
            if (n_geo->size() <= 8) { target->c2 = new BVH(*n_geo); delete n_geo; }
            else { target->c2 = new BVH(); packets.push({target->c2, n_geo}); }
            delete geo;
            packets.pop();
        }
        return i;
    }
};

// ---- packaged scene geometry --------------------------------------------
struct PackagedGeometry {
    std::vector<BVH_AVX> avx_bvh;
    std::vector<PTri_AVX> avx_tri_data;
    std::vector<PackagedTri> tri_data;
    double build_seconds = 0;
    bool simd = false;                     // casts through sro_trace_avx2.h (AVX2 intrinsics as the reference writes them) instead of the scalar loops
    // NOT part of the reference: the product's kernels skip a leaf whose padded tight box the ray misses (the
    // reference grows leaf boxes from the origin, quirk Q1, so most leaf visits cannot hit anything). The oracle
    // keeps the reference's traversal and only COUNTS the packet tests such a filter would save, with the same
    // boxes and the same arithmetic, so the GPU's packet-test counter can be checked exactly.
    // [node][child][min.xyz, max.xyz] = the planes the product stores per child (srx_scene.hip): the reference's,
    // except that a leaf box containing the origin on every axis has its planes at 0 replaced by the tight bound of
    // the leaf's triangles (from the packets, clamped into the reference box)
    std::vector<float> filter_boxes;
    std::vector<uint32_t> filter_cones;    // [node][child]: the leaf's normal-cone word (see leaf_cone_word), 0 = no guard needed
    float filter_pad = 0.0f;
    bool filter_enabled = true;            // mirrors SRX_LEAF_FILTER=0 of the product
    static constexpr float FILTER_PAD_REL = 5.0e-4f;   // pad = this x the largest |coordinate difference| between the ray's origin and the leaf's box ...
    static constexpr float FILTER_PAD_ABS = 4.0e-6f;   // ... + this x (scene size + largest |origin coordinate|): rounding of the slab test itself
    static constexpr uint32_t CONE_NEVER = 0xFF000000u;

    // NOT part of the reference. The guard word of a leaf: when a ray is closer than asin(kappa / sin(gamma)) to the
    // plane of a triangle (gamma = angle between its edges), the reference's fp32 Moller-Trumbore (Primitives.h:159-202)
    // divides by a det that is mostly rounding error and may accept a "hit" far outside the triangle, so a leaf may
    // only be skipped for rays that keep that distance from the planes of ALL its triangles. Stored as a cone around
    // the mean normal: bytes 0-2 = axis a (int8, largest component +-127), byte 3 = h: the leaf may be skipped iff
    // |a . dq| >= 127 h with dq = round(127 d / |d|); h = 255 never. Same arithmetic, in double, as srx_scene.hip.
    uint32_t leaf_cone_word(uint32_t first, uint32_t count) const {
        const double kappa = 1.0 / 80.0;
        struct V { double x, y, z; };
        std::vector<V> nh; std::vector<double> tau;
        V sum = {0, 0, 0};
        for (uint32_t p = first; p < first + count; p++) {
            const PTri_AVX& P = avx_tri_data[p];
            for (int k = 0; k < 8 && k < (int)P.size; k++) {
                const double ax = P.p1p2X[k], ay = P.p1p2Y[k], az = P.p1p2Z[k], bx = P.p1p3X[k], by = P.p1p3Y[k], bz = P.p1p3Z[k];
                const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
                const double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
                const double ll = std::sqrt(ax * ax + ay * ay + az * az) * std::sqrt(bx * bx + by * by + bz * bz);
                if (!(ll >= 1e-16) || !(ll <= 1e16) || !(nn > 0.0)) return CONE_NEVER;
                const double t = kappa / (nn / ll);
                if (!(t < 1.0)) return CONE_NEVER;
                V n = {nx / nn, ny / nn, nz / nn};
                if (!nh.empty() && n.x * nh[0].x + n.y * nh[0].y + n.z * nh[0].z < 0.0) { n.x = -n.x; n.y = -n.y; n.z = -n.z; }
                nh.push_back(n); tau.push_back(t);
                sum.x += n.x; sum.y += n.y; sum.z += n.z;
            }
        }
        if (nh.empty()) return 0u;
        const double m = std::max(std::fabs(sum.x), std::max(std::fabs(sum.y), std::fabs(sum.z)));
        if (!(m > 1e-6)) return CONE_NEVER;
        const double a[3] = {std::floor(sum.x / m * 127.0 + 0.5), std::floor(sum.y / m * 127.0 + 0.5), std::floor(sum.z / m * 127.0 + 0.5)};
        const double A = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        double thr = 0.0;
        for (size_t k = 0; k < nh.size(); k++) {
            const double c = std::fabs(nh[k].x * a[0] + nh[k].y * a[1] + nh[k].z * a[2]) / A;
            if (!(tau[k] < c)) return CONE_NEVER;
            const double s = std::sqrt(std::max(0.0, 1.0 - c * c));
            thr = std::max(thr, tau[k] * c + std::sqrt(1.0 - tau[k] * tau[k]) * s);
        }
        const double h = std::ceil(thr * A + (std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2])) / 254.0 + 0.01);
        if (!(h <= 254.0)) return CONE_NEVER;
        return ((uint32_t)(int)a[0] & 255u) | (((uint32_t)(int)a[1] & 255u) << 8) | (((uint32_t)(int)a[2] & 255u) << 16) | ((uint32_t)(int)h << 24);
    }

    void build_filter_boxes() {
        auto bounds = [&](uint32_t first, uint32_t count, float* mn, float* mx) {
            for (int ax = 0; ax < 3; ax++) { mn[ax] = 3.0e38f; mx[ax] = -3.0e38f; }
            for (uint32_t p = first; p < first + count; p++) {
                const PTri_AVX& P = avx_tri_data[p];
                const float* p1[3] = {P.p1X, P.p1Y, P.p1Z};
                const float* e1[3] = {P.p1p2X, P.p1p2Y, P.p1p2Z};
                const float* e2[3] = {P.p1p3X, P.p1p3Y, P.p1p3Z};
                for (int k = 0; k < 8 && k < (int)P.size; k++)
                    for (int ax = 0; ax < 3; ax++) {
                        const float v[3] = {p1[ax][k], p1[ax][k] + e1[ax][k], p1[ax][k] + e2[ax][k]};
                        for (float c : v) { mn[ax] = std::min(mn[ax], c); mx[ax] = std::max(mx[ax], c); }
                    }
            }
        };
        float scale = 1e-3f;
        if (!avx_tri_data.empty()) {
            float mn[3], mx[3];
            bounds(0, (uint32_t)avx_tri_data.size(), mn, mx);
            for (int ax = 0; ax < 3; ax++)
                if (mx[ax] >= mn[ax]) scale = std::max(scale, std::max(mx[ax] - mn[ax], std::max(std::fabs(mn[ax]), std::fabs(mx[ax]))));
        }
        filter_pad = FILTER_PAD_ABS * scale;
        filter_boxes.assign(avx_bvh.size() * 48, 0.0f);
        filter_cones.assign(avx_bvh.size() * 8, 0u);
        for (size_t i = 0; i < avx_bvh.size(); i++)
            for (int k = 0; k < 8; k++)
                if (avx_bvh[i].leaf_size[k] > 0) {
                    const BVH_AVX& n = avx_bvh[i];
                    const float rmin[3] = {n.minX[k], n.minY[k], n.minZ[k]}, rmax[3] = {n.maxX[k], n.maxY[k], n.maxZ[k]};
                    float* f = &filter_boxes[i * 48 + k * 6];
                    bool ext = true;
                    for (int ax = 0; ax < 3; ax++) { f[ax] = rmin[ax]; f[3 + ax] = rmax[ax]; ext = ext && rmin[ax] <= 0.0f && rmax[ax] >= 0.0f; }
                    if (!ext) continue;
                    float mn[3], mx[3];
                    bounds(n.indexes[k], n.leaf_size[k], mn, mx);
                    for (int ax = 0; ax < 3; ax++) {
                        if (rmin[ax] == 0.0f && rmax[ax] == 0.0f) { f[ax] = 0.0f; f[3 + ax] = 0.0f; }
                        else if (rmin[ax] == 0.0f) f[ax] = std::min(std::max(mn[ax], 0.0f), rmax[ax]);
                        else if (rmax[ax] == 0.0f) f[3 + ax] = std::max(std::min(mx[ax], 0.0f), rmin[ax]);
                    }
                    filter_cones[i * 8 + k] = leaf_cone_word(n.indexes[k], n.leaf_size[k]);
                }
    }
};

// BVH.h:989-1077
static inline void collapse(BVH* top, std::vector<BVH_AVX>& vec, std::vector<PackagedTri>& tri_vec) {
    typedef std::pair<BVH*, std::pair<int, int>> Item;
    std::vector<Item> current, next;
    int accumulated_index = 0;
    current.push_back(Item(top, std::pair<int, int>(-1, -1)));
    auto comparer = [](BVH* b1, BVH* b2) {                                    // :1002-1007
        return surface_area(b1->max, b1->min) * b1->count() > surface_area(b2->max, b2->min) * b2->count();
    };
    while (current.size() > 0) {
        for (size_t ci = 0; ci < current.size(); ci++) {
            Item selection = current.at(ci);
            std::vector<BVH*> batch, has_tris;
            batch.push_back(selection.first);
            while (true) {
                while (batch.size() + has_tris.size() < 8) {
                    BVH* at_index = batch[0];
                    if (at_index->c1 != nullptr) {
                        batch.push_back(at_index->c1);
                        batch.push_back(at_index->c2);
                        batch.erase(batch.begin());
                    } else {
                        has_tris.push_back(at_index);
                        batch.erase(batch.begin());
                        if (batch.size() == 0) break;
                    }
                    std::sort(batch.begin(), batch.end(), comparer);
                }
                if (has_tris.size() + batch.size() >= 8 || batch.size() == 0) {
                    if (has_tris.size() + batch.size() > 8) throw std::runtime_error("AVX BVH over allocated");
                    for (BVH* h : has_tris) batch.push_back(h);
                    break;
                }
            }
            int parent_index = selection.second.first;
            int point_back_index = selection.second.second;
            BVH_AVX A;                                                        // ctor :882-898
            std::memset((void*)&A, 0, sizeof(A));
            for (size_t i = 0; i < batch.size(); i++) {
                A.maxX[i] = batch[i]->max.X; A.maxY[i] = batch[i]->max.Y; A.maxZ[i] = batch[i]->max.Z;
                A.minX[i] = batch[i]->min.X; A.minY[i] = batch[i]->min.Y; A.minZ[i] = batch[i]->min.Z;
                A.leaf_size[i] = (uint32_t)batch[i]->elements.size();
            }
            A.parent_index = parent_index;
            A.self_index = point_back_index;
            for (size_t i = 0; i < batch.size(); i++) {
                if (A.leaf_size[i] > 0) {
                    A.indexes[i] = (uint32_t)tri_vec.size();
                    for (uint32_t j = 0; j < A.leaf_size[i]; j++) tri_vec.push_back(batch[i]->elements[j]->pack());
                } else {
                    next.push_back(Item(batch[i], std::pair<int, int>(accumulated_index, (int)i)));
                }
            }
            vec.push_back(A);
            if (parent_index != -1) vec[parent_index].indexes[point_back_index] = (uint32_t)accumulated_index;
            accumulated_index++;
        }
        current = next;
        next.clear();
    }
}

// Scene.h:90-179 (geometry part)
static inline void package(const std::vector<Tri>& tris, PackagedGeometry& PS) {
    BVH* bvh = new BVH();
    std::vector<const Tri*> ptrs;
    ptrs.reserve(tris.size());
    for (const Tri& T : tris) ptrs.push_back(&T);
    bvh->construct(ptrs);
    PS.tri_data.reserve(tris.size() + 1);
    collapse(bvh, PS.avx_bvh, PS.tri_data);
    delete bvh;
    // packet regroup :142-173
    std::map<int, std::pair<size_t, int>> order;
    for (size_t i = 0; i < PS.avx_bvh.size(); i++)
        for (int k = 0; k < 8; k++)
            if (PS.avx_bvh[i].leaf_size[k] > 0) order[(int)PS.avx_bvh[i].indexes[k]] = {i, k};
    for (auto& it : order) {
        int index = it.first;
        BVH_AVX& current = PS.avx_bvh[it.second.first];
        int sel = it.second.second;
        int leaf_size = (int)current.leaf_size[sel];
        current.indexes[sel] = (uint32_t)PS.avx_tri_data.size();
        int npk = (int)std::ceil(((float)leaf_size) / 8.0);
        current.leaf_size[sel] = (uint32_t)npk;
        for (int i = 0; i < npk; i++) {
            PTri_AVX P;                                                       // Primitives.h:135-155
            std::memset((void*)&P, 0, sizeof(P));
            int n = 0;
            for (int j = 0; j < 8 && i * 8 + j < leaf_size; j++) {
                const PackagedTri& t = PS.tri_data[index + 8 * i + j];
                P.p1X[j] = t.p1.X; P.p1Y[j] = t.p1.Y; P.p1Z[j] = t.p1.Z;
                P.p1p2X[j] = t.p1p2.X; P.p1p2Y[j] = t.p1p2.Y; P.p1p2Z[j] = t.p1p2.Z;
                P.p1p3X[j] = t.p1p3.X; P.p1p3Y[j] = t.p1p3.Y; P.p1p3Z[j] = t.p1p3.Z;
                P.origin_offset[j] = t.origin_offset;
                P.materials[j] = t.material;
                n++;
            }
            P.size = (int8_t)n;
            P.index = index + 8 * i;
            PS.avx_tri_data.push_back(P);
        }
    }
}

}  // namespace sro
