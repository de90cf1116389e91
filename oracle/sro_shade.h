// oracle/sro_shade.h — TEST INFRASTRUCTURE ONLY (the parity oracle).
//
// CPU restatement of the shading half of the hot path: texture/parameter
// fetch, MaterialSample BSDF math, the branched process_ray recursion with
// direct-light sampling and Russian roulette, the primary-ray camera and the
// running-mean accumulation contract. float/double promotions follow the
// reference expression by expression (a literal like 0.999 is a double there).
// PARITY UNPINNED (no upstream tests; see sro_math.h header and DESIGN.md).
#pragma once
#include "sro_trace.h"
#include "sro_trace_avx2.h"

namespace sro {

// ---------------------------------------------------------------- textures
// Texture.h:33-280 with USE_CHAR 1, USE_MORTON 0: RGB8 texels, row-major.
struct Texture {
    const uint8_t* data = nullptr;
    int resolution_x = 0, resolution_y = 0;
    XY scale = XY(1, 1);
    bool use_normal_transform = false;
    double U_multiplier = 0, V_multiplier = 0;
    void prep() {                                                            // Texture.h:227-230
        U_multiplier = resolution_x * scale.X;
        V_multiplier = resolution_y * scale.Y;
    }
    size_t address(int lx, int ly) const {                                   // lookup :245-252
        int x = lx % resolution_x;
        int y = ly % resolution_y;
        x = (x < 0) * resolution_x + x;
        y = (y < 0) * resolution_y + y;
        return (size_t)((unsigned)y * (unsigned)resolution_x + (unsigned)x);
    }
    void uv_to_xy(double U, double V, int& px, int& py) const {              // getUV :253-258
        V = 1 - V;
        px = (int)(U_multiplier * U);
        py = (int)(V_multiplier * V);
    }
    XYZ getUV_XYZ(double U, double V) const {                                // fetch_char_XYZ :147-155
        int px, py; uv_to_xy(U, V, px, py);
        size_t a = address(px, py);
        unsigned char r = data[a * 3 + 0], g = data[a * 3 + 1], b = data[a * 3 + 2];
        if (use_normal_transform) return XYZ((r - 128) / 127.0f, (g - 128) / 127.0f, (b - 128) / 127.0f);
        return XYZ(r / 255.0f, g / 255.0f, b / 255.0f);
    }
    float getUV_float(double U, double V) const {                            // fetch_char_float :156-159
        int px, py; uv_to_xy(U, V, px, py);
        size_t a = address(px, py);
        return data[a * 3 + 0] / 255.0f;
    }
};

// Materials.h:317-358 as plain data: each parameter is a constant or a texture id (-1 = constant).
struct Material {
    XYZ color = XYZ(0);          int color_tex = -1;
    float roughness = 0.1f;      int roughness_tex = -1;
    float metallic = 0;          int metallic_tex = -1;
    float specular = 0;          int specular_tex = -1;
    XYZ emissive = XYZ(0);       int emissive_tex = -1;
    XYZ normal = XYZ(0);         int normal_tex = -1;
    float transmission = 0;      int transmission_tex = -1;
    float IOR = 1;               int IOR_tex = -1;
    bool use_normals = false;
};

struct SunLight { XYZ rotation; XYZ emission; };                            // Primitives.h:22-51

// ---------------------------------------------------------------- MaterialSample
// Materials.h:13-315 (only the members the path reads)
struct MaterialSample {
    XYZ color; float roughness, metallic, specularFactor; XYZ emissive; XYZ normalPertubation;
    float transmission, IOR;
    XYZ normal, out, in, half;
    float n1, n2, R0; XYZ R0C;
    float i_metal, i_roughness, sigma;
    float a = 0, a_2 = 0;
    float dot_NO, dot_NI, dot_NH, dot_HI, dot_HO;
    float Ti, To, dot_AD;
    const float* arccos_lookup = nullptr;

    static float poly_acos(float x) {                                        // Materials.h:111-124
        float negate = float(x < 0);
        x = std::fabs(x);
        float ret = -0.0187293;
        ret = ret * x;
        ret = ret + 0.0742610;
        ret = ret * x;
        ret = ret - 0.2121144;
        ret = ret * x;
        ret = ret + 1.5707288;
        ret = ret * std::sqrt(1.0 - x);
        ret = ret - 2 * negate * ret;
        return negate * 3.14159265358979 + ret;
    }
    void config(XYZ _out, XYZ _normal, float _n1, float _n2) {              // Materials.h:77-110
        out = _out;
        normal = _normal;
        dot_NO = std::min(XYZ::dot(out, normal), 1.0f);
        n1 = _n1; n2 = _n2;
        i_metal = 1 * metallic;
        i_roughness = roughness * 0.999 + 0.001;
        sigma = roughness * 0.3;
        To = poly_acos(dot_NO);
        float spec = std::pow((n1 - n2) / (n1 + n2), 2);                    // get_R0 :144-147
        R0 = ((1 - i_metal) * spec + i_metal);
        R0C = XYZ::linear_mix(metallic, XYZ(R0), i_metal * color);          // get_R0C :148-150
        a = i_roughness;
        a_2 = a * a;
    }
    void set_input(XYZ _in) {                                                // Materials.h:125-143
        in = _in;
        half = XYZ::normalize((out + in) / 2);
        dot_NI = std::min(1.0f, XYZ::dot(in, normal));
        dot_NH = std::min(1.0f, XYZ::dot(half, normal));
        dot_HI = std::min(1.0f, XYZ::dot(half, in));
        dot_HO = std::min(1.0f, XYZ::dot(half, out));
        Ti = Lacos(arccos_lookup, dot_NI);
        XYZ projected_in = XYZ::normalize(in - normal * XYZ::dot(in, normal));
        XYZ projected_out = XYZ::normalize(out - normal * XYZ::dot(out, normal));
        dot_AD = XYZ::dot(projected_in, projected_out);
    }
    XYZ fast_fresnel(float d) const {                                        // :174-177 (quirk Q4: 1 - R0C adds)
        float g = 1 - d;
        return R0C + float_minus_xyz(1, R0C) * (g * g * g * g * g);
    }
    float fast_fresnel(const XYZ& nrm, const XYZ& vec) const {               // :178-181
        float g = 1 - XYZ::dot(nrm, vec);
        return R0 + (1 - R0) * (g * g * g * g * g);
    }
    float normal_dist_ggx() const {                                          // :190-194
        float c = dot_NH;
        float s = std::sqrt(1 - dot_NH * dot_NH);
        return std::pow(1 + std::pow(s / c, 2) / a_2, -2);
    }
    float G1(const XYZ& vec) const {                                         // :220-225
        float c = std::min(XYZ::dot(vec, normal), 0.99f);
        float s_2 = 1 - c * c;
        float t_2 = s_2 / (c * c);
        return 1 / (1 + (-1 + std::sqrt(1 + a * a * t_2)) / 2);
    }
    float G_GGX() const { return G1(out) * G1(in); }                         // :226-229
    XYZ oren_nayar_diffuse_BRDF() const {                                    // :236-253
        float s_2 = sigma * sigma;
        float aa = std::max(Ti, To);
        float bb = std::min(Ti, To);
        float A = 1 - 0.5 * s_2 / (s_2 + 0.33);
        float B = 0.45 * s_2 / (s_2 + 0.09) * (full_sin(aa) - ((dot_AD >= 0) * std::pow(2 * bb / PI_D, 3)));
        float C = 0.125 * s_2 / (s_2 + 0.09) * std::pow(4 * aa * bb / PI_D, 2);
        XYZ modifier = color / (float)PI_D;
        float L1 = (A + B * std::max(0.0f, dot_AD) * vtan(bb) + C * (1 - std::fabs(dot_AD)) * vtan((aa + bb) / 2));
        float L2 = 0.17 * s_2 / (s_2 + 0.13) * (1 - dot_AD) * std::pow(2 * bb / PI_D, 2);
        return modifier * L1 + color * modifier * L2;
    }
    XYZ diffuse_BRDF() const { return oren_nayar_diffuse_BRDF(); }           // :254-257
    XYZ specular_BRDF() const {                                              // :258-266 (use_fresnel = true)
        if (dot_NH < 0 || dot_NO < 0 || dot_NI < 0 || dot_HI < 0 || dot_HO < 0) return XYZ(0, 0, 0);
        XYZ fresnel = (R0C * 0.0f) + 1.0f * fast_fresnel(dot_HO);
        float geo = G_GGX();
        float normal_dist = normal_dist_ggx();
        float divisor = 4 * dot_NO * dot_NI;
        return geo * normal_dist * fresnel / divisor;
    }
    XYZ PDF_specular_BRDF() const {                                          // :267-272
        if (dot_NH < 0 || dot_NO < 0 || dot_NI < 0 || dot_HI < 0 || dot_HO < 0) return XYZ(0, 0, 0);
        XYZ tint = XYZ::linear_mix(metallic, XYZ(1, 1, 1), color);
        XYZ weight = XYZ(G_GGX() / G1(in));
        return tint * weight;
    }
    XYZ fast_BRDF_co(bool apply_scalar) const {                              // :273-280
        XYZ specular_return = specular_BRDF();
        XYZ light_remainder = XYZ(1) - specular_return;
        XYZ diffuse_return = light_remainder * diffuse_BRDF();
        return (specular_return + diffuse_return) * ((apply_scalar) ? dot_NI : 1);
    }
};

// ---------------------------------------------------------------- scene + engine
struct Scene {
    PackagedGeometry geo;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<SunLight> lights;
    Tables tables;
    short max_generations = 10;            // Scene.h:27-30
    short monte_carlo_max = 32;
    float monte_carlo_modifier = 1.0 / 4;
};

struct PackagedRay {                        // Ray.h:7-25 (IOR stack omitted: quirk Q5 makes it unread)
    XYZ position, slope;
    signed char generation = 0;
    XYZ coefficient = XYZ(1, 1, 1);
    uint64_t key = 0;                       // keyed-RNG mode only
};

struct Diagnostics {                        // Coredefs.h:17-23 + oracle-side extras
    long long rays_processed = 0;
    long long diffuses_cast = 0;
    long long shadows_cast = 0;
    TraceCounters trace;                    // closest-hit casts
    TraceCounters shadow_trace;             // shadow casts
    long long hits = 0, texels = 0, pdf_lookups = 0;
};

struct RayEngine {
    const Scene* data = nullptr;
    Rng* shared_gen = nullptr;              // reference mode: one sequential stream (commons.cpp:30)
    bool keyed = false;                     // wavefront mode: per-ray KeyedGen seeded from ray.key
    // parity ray sets: when set, a generation-0 call appends the child rays it queued (Core.h:573-588) and
    // returns without tracing them
    std::vector<PackagedRay>* capture_first_bounce = nullptr;

    MaterialSample sample_UV(const Material& m, float U, float V, Diagnostics& st) const {  // Materials.h:346-357
        MaterialSample s;
        const std::vector<Texture>& tx = data->textures;
        auto f = [&](float c, int t) { if (t < 0) return c; st.texels++; return tx[t].getUV_float(U, V); };
        auto v = [&](const XYZ& c, int t) { if (t < 0) return c; st.texels++; return tx[t].getUV_XYZ(U, V); };
        s.color = v(m.color, m.color_tex);
        s.roughness = f(m.roughness, m.roughness_tex);
        s.metallic = f(m.metallic, m.metallic_tex);
        s.specularFactor = f(m.specular, m.specular_tex);
        s.transmission = f(m.transmission, m.transmission_tex);
        s.IOR = f(m.IOR, m.IOR_tex);
        s.emissive = v(m.emissive, m.emissive_tex);
        s.normalPertubation = v(m.normal, m.normal_tex);
        s.arccos_lookup = data->tables.arccos_lookup.data();
        return s;
    }

    // Core.h:351-358 + SunLight::intersects Primitives.h:44-50
    bool execute_lighting_cast(XYZ position, const SunLight& L, Diagnostics& st) const {
        XYZ slope = L.rotation;
        CastResults res = execute_ray_cast(data->geo, position, slope, &st.shadow_trace);
        st.shadows_cast++;
        return res.distance > 9999;
    }

    // Core.h:359-601
    XYZ process_ray(Diagnostics& stats, PackagedRay& ray_data) const {
        stats.rays_processed++;
        KeyedGen local;
        Rng* g = shared_gen;
        if (keyed) { local.s = ray_data.key; g = &local; }

        XYZ orig = ray_data.position;
        CastResults results = execute_ray_cast(data->geo, ray_data.position, ray_data.slope, &stats.trace);
        const PackagedTri& Tri = data->geo.tri_data[results.tri_index];
        results.texUV.X = results.TUV.Y * Tri.U_delta.X + results.TUV.Z * Tri.V_delta.X + Tri.UV_basis.X;
        results.texUV.Y = results.TUV.Y * Tri.U_delta.Y + results.TUV.Z * Tri.V_delta.Y + Tri.UV_basis.Y;
        if (results.material == 0xFFFFFFFFu) return XYZ(0);
        stats.hits++;
        const Material& parent_material = data->materials[results.material];
        MaterialSample material = sample_UV(parent_material, results.texUV.X, results.texUV.Y, stats);

        bool backfacing = XYZ::dot(Tri.normal, orig - Tri.origin_offset) < 0;
        XYZ geo_normal = backfacing ? -Tri.normal : Tri.normal;
        const float u = results.TUV.Y, v = results.TUV.Z;
        const float k = 1 - u - v;                                           // Primitives.h:95-106
        XYZ smoothed_normal_raw = XYZ::normalize(Tri.n1 * k + Tri.n2 * u + Tri.n3 * v);
        XYZ normal = (XYZ::dot(geo_normal, smoothed_normal_raw) < 0) ? -smoothed_normal_raw : smoothed_normal_raw;
        XYZ T = XYZ::normalize(Tri.t1 * k + Tri.t2 * u + Tri.t3 * v);
        XYZ BT = XYZ::normalize(Tri.b1 * k + Tri.b2 * u + Tri.b3 * v);
        if (parent_material.use_normals) {
            normal = material.normalPertubation.X * T + material.normalPertubation.Y * BT +
                     material.normalPertubation.Z * normal;
        }
        normal = XYZ::normalize(normal);

        XYZ aggregate = XYZ();
        aggregate += material.emissive;                                      // :426

        int bounce_count = (int)(data->monte_carlo_max * std::pow(data->monte_carlo_modifier, ray_data.generation)); // :428
        XYZ flipped_output = XYZ::flip(ray_data.slope);
        float n1 = 1.0003;                                                   // :437 (quirk Q5)
        float ior_u = n1 / material.IOR;

        material.config(flipped_output, normal, n1, material.IOR);           // :454

        for (const SunLight& L : data->lights) {                             // :457-471
            XYZ pos = ray_data.position + 0.001 * geo_normal;
            XYZ slope = L.rotation;
            material.set_input(slope);
            XYZ return_coefficient = material.fast_BRDF_co(true);
            if (execute_lighting_cast(pos, L, stats)) aggregate += return_coefficient * L.emission;
        }
        if (ray_data.generation >= data->max_generations) return aggregate;  // :473
        if (bounce_count < 1) bounce_count = 1;
        int ri = 0;
        if (XYZ::dot(flipped_output, normal) < 0) return aggregate;          // :489-491

        std::vector<PackagedRay> r_queue;
        r_queue.reserve(bounce_count);
        XYZ throughput = XYZ(1, 1, 1);
        const float* pdf_lookup = data->tables.pdf_lookup.data();

        for (int i = 0; i < bounce_count; i++) {                             // :498-591
            ri++;
            g->regen_pool();
            if (ri > bounce_count * 10 && i == 0) return aggregate;
            float factor = g->pool_rand(0);
            float c_theta = pdf::sample(pdf_lookup, factor, material.i_roughness);
            stats.pdf_lookups++;
            float azimuth = g->pool_rand(1) * 2.0f * fPI;
            float c_theta_sin = full_sin(c_theta + fPI);
            float c_theta_cos = full_cos(c_theta + fPI);
            XYZ MN_1 = XYZ(-c_theta_sin * full_cos(azimuth), -c_theta_sin * full_sin(azimuth), -c_theta_cos);
            MN_1 = applyTBNMatrix(MN_1, T, BT, normal);
            XYZ MN_2 = XYZ::reflect(MN_1, normal);
            float choice_prob = XYZ::cdot(flipped_output, MN_2) /
                                (XYZ::cdot(flipped_output, MN_1) + XYZ::cdot(flipped_output, MN_2));
            float choice_factor = g->pool_rand(2);
            XYZ MN = MN_1;
            if (choice_factor < choice_prob) MN = MN_2;
            float ref_prob = material.fast_fresnel(MN, flipped_output);
            float ref_factor = g->pool_rand(3);
            XYZ slope;
            XYZ return_coefficient = throughput;
            bool is_transmissive = false;
            if (ref_factor < ref_prob) {
                slope = XYZ::reflect(flipped_output, MN);
                material.set_input(slope);
                return_coefficient *= material.PDF_specular_BRDF();
            } else {
                float trans_factor = g->pool_rand(4);
                if (trans_factor < material.transmission) {
                    float dot_HI = XYZ::dot(MN, flipped_output);
                    float f1 = std::sqrt(1 - ior_u * ior_u * (1 - dot_HI * dot_HI));
                    slope = -(f1 * MN + ior_u * (flipped_output - dot_HI * MN));
                    is_transmissive = true;
                } else {
                    // VecLib.cpp:372-382 via biased_random_hemi :408-410
                    const float r1 = g->fRand(0, 1);
                    const float r2 = g->fRand(0, 1);
                    const float r = std::sqrt(r1);
                    const float theta = 2 * PI_D * r2;
                    const float hx = r * full_cos(theta);
                    const float hy = r * full_sin(theta);
                    XYZ hemi(hx, std::sqrt(1 - r1), hy);
                    slope = XYZ(hemi.X, hemi.Z, hemi.Y);
                    slope = applyTBNMatrix(slope, T, BT, normal);
                    material.set_input(slope);
                    return_coefficient *= material.diffuse_BRDF();
                }
            }
            bool t_flip = is_transmissive;
            if (t_flip != (XYZ::dot(slope, normal) < 0)) { throughput = return_coefficient; i--; continue; }
            if (t_flip != (XYZ::dot(slope, geo_normal) < 0)) { throughput = XYZ(1, 1, 1); i--; continue; }
            if (return_coefficient.X < 0 || return_coefficient.Y < 0 || return_coefficient.Z < 0) {
                throughput = XYZ(1, 1, 1); i--; continue;
            }
            XYZ global_return = return_coefficient;
            float russian_p = std::max(global_return.X, std::max(global_return.Y, global_return.Z));
            float russian_factor = g->pool_rand(5);
            if (russian_factor > russian_p) { throughput = XYZ(1, 1, 1); continue; }
            return_coefficient = return_coefficient / russian_p;
            PackagedRay ray;
            ray.position = ray_data.position + (float)((is_transmissive ? -1 : 1) * 0.0001) * geo_normal;
            ray.slope = slope;
            ray.generation = (signed char)(ray_data.generation + 1);
            ray.coefficient = return_coefficient;
            if (keyed) ray.key = local.child_key();
            r_queue.push_back(ray);
            throughput = XYZ(1, 1, 1);
        }
        if (capture_first_bounce && ray_data.generation == 0) {
            capture_first_bounce->insert(capture_first_bounce->end(), r_queue.begin(), r_queue.end());
            return aggregate;
        }
        for (size_t i = 0; i < r_queue.size(); i++) {                        // :592-598
            PackagedRay& ray = r_queue[i];
            ray.coefficient = ray.coefficient / bounce_count;
            stats.diffuses_cast++;
            XYZ returned_light = process_ray(stats, ray);
            aggregate += ray.coefficient * returned_light;
        }
        return aggregate;
    }
};

// ---------------------------------------------------------------- camera
// Camera.h:32-140 (RectLens + Camera)
struct Camera {
    XYZ position;
    Quat rotation;
    XYZ focal_position;
    float width = 1, height = 1;
    int res_x = 0, res_y = 0, subdiv = 1;
    float pixel_width = 0, pixel_height = 0, subdiv_width = 0, subdiv_height = 0, offset_x = 0, offset_y = 0;
    std::vector<XY> subdiv_offsets;
    void prep(int rx, int ry, int subdivision_count) {                       // :50-85
        res_x = rx; res_y = ry; subdiv = subdivision_count;
        float partial_width = width / rx;
        float partial_height = height / ry;
        pixel_width = partial_width; pixel_height = partial_height;
        offset_x = -width / 2; offset_y = -height / 2;
        subdiv_width = partial_width / subdivision_count;
        subdiv_height = partial_height / subdivision_count;
        subdiv_offsets.clear();
        for (int y = 0; y < subdivision_count; y++)
            for (int x = 0; x < subdivision_count; x++)
                subdiv_offsets.push_back(XY(subdiv_width * (x + 0.5), subdiv_height * (y + 0.5)));
    }
    XY lens_output(int px, int py) const {                                   // outputs[y][x] :72-83
        return XY(offset_x + pixel_width * px, offset_y + pixel_height * py);
    }
    XYZ random_slope_at(int px, int py, int subindex, Rng& g) const {        // :91-97,138-140
        XY offset = lens_output(px, py);
        XY subdiv_offset = subdiv_offsets[subindex] - XY(subdiv_width, subdiv_height) * 0.5f;
        // Camera.h:95 draws both jitters inside one constructor call (order unspecified in C++);
        // the oracle fixes it left to right, as clang evaluates it.
        float jx = g.fRand(0, subdiv_width);
        float jy = g.fRand(0, subdiv_height);
        XY pos = offset + subdiv_offset + XY(jx, jy);
        XYZ lens(pos.X, 0, pos.Y);
        return applyRotation(XYZ::slope(XYZ(0, 0, 0), lens + focal_position), rotation);
    }
};

}  // namespace sro
