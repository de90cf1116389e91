// raytracing_amd/csrc/srx_shade.hip — ray generation, shade/bounce and accumulation kernels (gfx950).
//
// k_shade replaces the body of RayEngine::process_ray (Core.h:359-601) for one generation of a
// wavefront queue: one thread per traced ray reads its hit record, fetches the 160-B shading triangle,
// samples the material (Materials.h:346-357, Texture.h:245-258), adds the emissive term, queues one
// shadow ray per sun light with the direct-light coefficient (Core.h:457-471), and runs the bounce loop
// (GGX microfacet normal from the inverse-CDF table, Fresnel choice, transmission, cosine hemisphere with
// Oren-Nayar, rejection rules, Russian roulette — Core.h:498-591). The reference recurses; here each
// surviving child carries the product of coefficients down the path (its pixel contribution is
// throughput x radiance) and is appended to the next generation's queue with one atomic per wave
// (ballot + prefix count): that append IS the queue compaction between bounces.
// float/double promotions follow the reference expression by expression.
#include "srx_kernels.h"

// ---------------------------------------------------------------- material sample
struct MatSample {
    f3 color; float roughness, metallic; f3 emissive; f3 normalPert; float transmission, IOR;
    f3 normal, out, in;
    float R0; f3 R0C;
    float i_roughness, sigma, a, a_2;
    float dot_NO, dot_NI, dot_NH, dot_HI, dot_HO, Ti, To, dot_AD;

    __device__ static float poly_acos(float x) {                       // Materials.h:111-124
        float negate = float(x < 0);
        x = fabsf(x);
        float ret = -0.0187293;
        ret = ret * x;
        ret = ret + 0.0742610;
        ret = ret * x;
        ret = ret - 0.2121144;
        ret = ret * x;
        ret = ret + 1.5707288;
        ret = ret * sqrt(1.0 - x);
        ret = ret - 2 * negate * ret;
        return negate * 3.14159265358979 + ret;
    }
    __device__ void config(f3 _out, f3 _normal, float n1, float n2) {  // Materials.h:77-110
        out = _out;
        normal = _normal;
        dot_NO = std_min(dot3(out, normal), 1.0f);
        const float i_metal = 1 * metallic;
        i_roughness = roughness * 0.999 + 0.001;
        sigma = roughness * 0.3;
        To = poly_acos(dot_NO);
        const double q = (double)((n1 - n2) / (n1 + n2));
        const float spec = (float)(q * q);                             // pow(x, 2), :145
        R0 = ((1 - i_metal) * spec + i_metal);
        R0C = linear_mix3(metallic, f3(R0), i_metal * color);
        a = i_roughness;
        a_2 = a * a;
    }
    __device__ void set_input(f3 _in, const float* arccos_lookup) {    // Materials.h:125-143
        in = _in;
        const f3 half = normalize3((out + in) / 2.0f);
        dot_NI = std_min(1.0f, dot3(in, normal));
        dot_NH = std_min(1.0f, dot3(half, normal));
        dot_HI = std_min(1.0f, dot3(half, in));
        dot_HO = std_min(1.0f, dot3(half, out));
        const int index = ((int)((dot_NI + 1) / 2.0 * 2048)) & 2047;   // VecLib::Lacos, VecLib.cpp:360-363
        Ti = arccos_lookup[index];
        const f3 projected_in = normalize3(in - normal * dot3(in, normal));
        const f3 projected_out = normalize3(out - normal * dot3(out, normal));
        dot_AD = dot3(projected_in, projected_out);
    }
    __device__ f3 fast_fresnel_v(float d) const {                      // :174-177; `1 - R0C` ADDS (quirk Q4, XYZ.cpp:431)
        const float g = 1 - d;
        const f3 one_minus = f3(R0C.x + 1.0f, R0C.y + 1.0f, R0C.z + 1.0f);
        return R0C + one_minus * (g * g * g * g * g);
    }
    __device__ float fast_fresnel_s(f3 nrm, f3 vec) const {            // :178-181
        const float g = 1 - dot3(nrm, vec);
        return R0 + (1 - R0) * (g * g * g * g * g);
    }
    __device__ float normal_dist_ggx() const {                         // :190-194
        const float c = dot_NH;
        const float s = sqrtf(1 - dot_NH * dot_NH);
        const double r = (double)(s / c);
        const double base = 1 + (r * r) / (double)a_2;
        return (float)(1.0 / (base * base));                           // pow(base, -2)
    }
    __device__ float G1(f3 vec) const {                                // :220-225
        const float c = std_min(dot3(vec, normal), 0.99f);
        const float s_2 = 1 - c * c;
        const float t_2 = s_2 / (c * c);
        return 1 / (1 + (-1 + sqrtf(1 + a * a * t_2)) / 2);
    }
    __device__ float G_GGX() const { return G1(out) * G1(in); }
    __device__ f3 oren_nayar() const {                                 // :236-253
        const float s_2 = sigma * sigma;
        const float aa = std_max(Ti, To);
        const float bb = std_min(Ti, To);
        const float A = 1 - 0.5 * s_2 / (s_2 + 0.33);
        const double b2pi = 2 * bb / SRX_PI;
        const float B = 0.45 * s_2 / (s_2 + 0.09) * (full_sin(aa) - ((dot_AD >= 0) * (b2pi * b2pi * b2pi)));
        const double ab = 4 * aa * bb / SRX_PI;
        const float C = 0.125 * s_2 / (s_2 + 0.09) * (ab * ab);
        const f3 modifier = color / (float)SRX_PI;
        const float L1 = (A + B * std_max(0.0f, dot_AD) * vtan(bb) + C * (1 - fabsf(dot_AD)) * vtan((aa + bb) / 2));
        const float L2 = 0.17 * s_2 / (s_2 + 0.13) * (1 - dot_AD) * (b2pi * b2pi);
        return modifier * L1 + color * modifier * L2;
    }
    __device__ f3 specular_BRDF() const {                              // :258-266
        if (dot_NH < 0 || dot_NO < 0 || dot_NI < 0 || dot_HI < 0 || dot_HO < 0) return f3(0, 0, 0);
        const f3 fresnel = (R0C * 0.0f) + 1.0f * fast_fresnel_v(dot_HO);
        const float geo = G_GGX();
        const float nd = normal_dist_ggx();
        const float divisor = 4 * dot_NO * dot_NI;
        return (geo * nd) * fresnel / divisor;
    }
    __device__ f3 PDF_specular_BRDF() const {                          // :267-272
        if (dot_NH < 0 || dot_NO < 0 || dot_NI < 0 || dot_HI < 0 || dot_HO < 0) return f3(0, 0, 0);
        const f3 tint = linear_mix3(metallic, f3(1, 1, 1), color);
        const f3 weight = f3(G_GGX() / G1(in));
        return tint * weight;
    }
    __device__ f3 fast_BRDF_co() const {                               // :273-280 with apply_scalar = true
        const f3 sr = specular_BRDF();
        const f3 remainder = f3(1.0f) - sr;
        const f3 dr = remainder * oren_nayar();
        return (sr + dr) * dot_NI;
    }
};

// Texture.h:245-258 + fetch_char_* :147-159 (V flipped, C-style wrap); same texel as the reference's row-major
// lookup, fetched from the Morton-tiled copy with one dword load.
__device__ __forceinline__ uint32_t spread5(uint32_t v) {     // abcde -> 0a0b0c0d0e
    v = (v | (v << 4)) & 0x10Fu;
    v = (v | (v << 2)) & 0x133u;
    v = (v | (v << 1)) & 0x155u;
    return v;
}
__device__ __forceinline__ uint32_t tex_texel(const GTexture& t, float U, float V) {
    double dU = U, dV = V;
    dV = 1 - dV;
    const int lx = (int)(t.U_multiplier * dU);
    const int ly = (int)(t.V_multiplier * dV);
    int x = lx % t.res_x;
    int y = ly % t.res_y;
    x = (x < 0) * t.res_x + x;
    y = (y < 0) * t.res_y + y;
    const uint32_t ux = (uint32_t)x, uy = (uint32_t)y;
    const size_t tile = (size_t)(uy >> 5) * (uint32_t)t.tiles_x + (ux >> 5);
    return t.data[tile * 1024u + (spread5(ux & 31u) | (spread5(uy & 31u) << 1))];
}
__device__ __forceinline__ f3 tex_xyz(const GTexture& t, float U, float V) {
    const uint32_t p = tex_texel(t, U, V);
    const int r = (int)(p & 0xFFu), g = (int)((p >> 8) & 0xFFu), b = (int)((p >> 16) & 0xFFu);
    if (t.normal_transform) return f3((r - 128) / 127.0f, (g - 128) / 127.0f, (b - 128) / 127.0f);
    return f3(r / 255.0f, g / 255.0f, b / 255.0f);
}
__device__ __forceinline__ float tex_float(const GTexture& t, float U, float V) {
    return (int)(tex_texel(t, U, V) & 0xFFu) / 255.0f;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}

// ---------------------------------------------------------------- shade
// Staging slots per wave for surviving children. A flush is one returning atomic on the queue counter plus the
// write-out, and the wave stalls for it: with 128 slots (flush above 64) the shade kernel took 50.1 ms per 4 passes, with
// 176 / 192 / 208 slots 41.4 / 41.1 / 40.9 ms, with 256 (LDS for 3 blocks per CU instead of 4) 43.6 ms.
#ifndef SRX_STAGE
#define SRX_STAGE 192
#endif
#ifndef SRX_SHADE_WAVES
#define SRX_SHADE_WAVES 4
#endif
// WAVES: waves per block. The block shares the two queue reservations (shadow rays at the top, the last children at the
// end), so its waves leave together, after the slowest of them; srx_launch_shade picks WAVES per generation.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_shade(GScene sc, ShadeArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool live = idx < (a.n_ptr ? min(*a.n_ptr, a.n) : a.n);
    __shared__ GRay s_stage[WAVES][SRX_STAGE];          // per-wave staging of surviving children (36 KB per block: 4 blocks = 16 waves per CU, what 118 VGPRs allow)
    __shared__ uint32_t s_wave_count[WAVES], s_wave_parked[WAVES];
    __shared__ uint32_t s_base[3];
    GRay* const stage = s_stage[threadIdx.x >> 6];
    uint32_t staged = 0;                      // wave-uniform

    GRay ray;
    GHit hit;
    hit.material = 0xFFFFFFFFu;
    const bool resumed = a.cont_in != nullptr;        // a later round: the launch's units are parked loops, not rays
    GCont rec;
    uint32_t slot = idx;
    if (live && resumed) { rec = a.cont_in[idx]; slot = rec.idx; }
    else if (live && a.slot_list) slot = a.slot_list[idx];
    if (live) { ray = a.rays[slot]; hit = a.hits[slot]; }
    const bool is_hit = live && hit.material != 0xFFFFFFFFu;

    uint32_t n_texels = 0, n_pdf = 0, n_children = 0;
    MatSample m;
    f3 T, BT, normal, geo_normal, position, flipped_output, thr;
    float ior_u = 1.0f;
    int bounce_count = 0;
    bool active = false;
    KeyedGen g;
    g.s = ((uint64_t)ray.key_hi << 32) | ray.key_lo;

    if (is_hit) {
        const f3 orig(ray.ox, ray.oy, ray.oz);
        const f3 slope(ray.dx, ray.dy, ray.dz);
        thr = f3(ray.cr, ray.cg, ray.cb);
        position = orig + (hit.t * slope) * 0.999f;                    // Core.h:327
        const GTri& tr = sc.tris[hit.tri_index];
        const float4 r0 = tr.r[0], r1 = tr.r[1], r2 = tr.r[2], r3 = tr.r[3], r4 = tr.r[4];
        const float4 r5 = tr.r[5], r6 = tr.r[6], r7 = tr.r[7], r8 = tr.r[8], r9 = tr.r[9];
        const float tu = hit.u, tv = hit.v;
        const float texU = tu * r0.z + tv * r1.x + r0.x;               // Core.h:365-366
        const float texV = tu * r0.w + tv * r1.y + r0.y;
        const f3 tri_normal(r1.w, r2.x, r2.y), origin_offset(r2.z, r2.w, r3.x);
        const f3 n1(r3.y, r3.z, r3.w), n2(r4.x, r4.y, r4.z), n3(r4.w, r5.x, r5.y);
        const f3 t1(r5.z, r5.w, r6.x), t2(r6.y, r6.z, r6.w), t3(r7.x, r7.y, r7.z);
        const f3 b1(r7.w, r8.x, r8.y), b2(r8.z, r8.w, r9.x), b3(r9.y, r9.z, r9.w);

        const GMaterial& pm = sc.materials[hit.material];              // Materials.h:346-357
        auto pf = [&](float c, int t) { if (t < 0) return c; n_texels++; return tex_float(sc.textures[t], texU, texV); };
        auto pv = [&](const float* c, int t) { if (t < 0) return f3(c[0], c[1], c[2]); n_texels++; return tex_xyz(sc.textures[t], texU, texV); };
        m.color = pv(pm.color, pm.color_tex);
        m.roughness = pf(pm.roughness, pm.roughness_tex);
        m.metallic = pf(pm.metallic, pm.metallic_tex);
        (void)pf(pm.specular, pm.specular_tex);
        m.transmission = pf(pm.transmission, pm.transmission_tex);
        m.IOR = pf(pm.ior, pm.ior_tex);
        m.emissive = pv(pm.emissive, pm.emissive_tex);
        m.normalPert = pv(pm.normal, pm.normal_tex);

        const bool backfacing = dot3(tri_normal, orig - origin_offset) < 0;    // Core.h:374-390
        geo_normal = backfacing ? -tri_normal : tri_normal;
        const float k = 1 - tu - tv;
        const f3 smoothed = normalize3(n1 * k + n2 * tu + n3 * tv);
        normal = (dot3(geo_normal, smoothed) < 0) ? -smoothed : smoothed;
        T = normalize3(t1 * k + t2 * tu + t3 * tv);
        BT = normalize3(b1 * k + b2 * tu + b3 * tv);
        if (pm.use_normals) normal = m.normalPert.x * T + m.normalPert.y * BT + m.normalPert.z * normal;
        normal = normalize3(normal);

        // emissive term (Core.h:426): contribution = throughput * emissive
        const f3 e = thr * m.emissive;
        if (!resumed && (e.x != 0.0f || e.y != 0.0f || e.z != 0.0f)) {
            float* dst = a.sample3 + 3 * (size_t)ray.pixel;
            atomicAdd(dst, e.x); atomicAdd(dst + 1, e.y); atomicAdd(dst + 2, e.z);
        }
        bounce_count = sc.bounce_count[a.generation];                  // Core.h:428
        flipped_output = -slope;
        const float n1_ior = 1.0003;                                   // Core.h:437 (quirk Q5)
        ior_u = n1_ior / m.IOR;
        m.config(flipped_output, normal, n1_ior, m.IOR);               // Core.h:454
    }

    // ---- direct light: one shadow ray per sun (Core.h:457-471). One reservation per BLOCK: a returning atomic
    // on one hot word per wave would serialise the chip (~88 of them per microsecond, chip-wide).
    const int wave = threadIdx.x >> 6;
    const unsigned long long hit_bal = __ballot(is_hit);
    if (!resumed) {                                   // (uniform over the launch)
    if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(hit_bal);
    __syncthreads();
    uint32_t hit_rank = (uint32_t)__popcll(hit_bal & lt_mask), block_hits = 0;
#pragma unroll
    for (int wv = 0; wv < WAVES; wv++) {
        const uint32_t c = s_wave_count[wv];
        if (wv < wave) hit_rank += c;
        block_hits += c;
    }
    if (threadIdx.x == 0 && sc.n_suns && block_hits) s_base[0] = atomicAdd(a.shadow_count, block_hits * sc.n_suns);
    __syncthreads();
    for (uint32_t s = 0; s < sc.n_suns; s++) {
        if (is_hit) {
            const f3 pos = position + 0.001f * geo_normal;
            const f3 ldir(sc.suns[s].dir[0], sc.suns[s].dir[1], sc.suns[s].dir[2]);
            m.set_input(ldir, sc.arccos_lookup);
            const f3 rc = m.fast_BRDF_co();
            const f3 contrib = thr * (rc * f3(sc.suns[s].emission[0], sc.suns[s].emission[1], sc.suns[s].emission[2]));
            GShadow q;
            q.ox = pos.x; q.oy = pos.y; q.oz = pos.z;
            q.pixel_sun = (ray.pixel & 0x00FFFFFFu) | (s << 24);
            q.cr = contrib.x; q.cg = contrib.y; q.cb = contrib.z; q.pad = 0.0f;
            a.shadow_q[s_base[0] + hit_rank * sc.n_suns + s] = q;
        }
    }
    }

    if (is_hit) {
        active = !(a.generation >= sc.max_generations);                // Core.h:473
        if (bounce_count < 1) bounce_count = 1;
        if (dot3(flipped_output, normal) < 0) active = false;          // Core.h:489-491
    }

    // ---- bounce loop (Core.h:498-591): every lane makes one attempt per trip; survivors are appended
    int i = 0, ri = 0;
    f3 throughput(1, 1, 1);
    if (resumed && is_hit) {
        i = (int)rec.i; ri = (int)rec.ri; throughput = f3(rec.tr, rec.tg, rec.tb); g.s = rec.s;
        n_texels = 0;                                 // counted by the first round
    }
    int trips = 0;
    while (__any(active)) {
        bool emit = false;
        GRay child;
        if (active) {
            ri++;
            g.regen_pool();
            if (ri > bounce_count * 10 && i == 0) {
                active = false;
            } else {
                const float factor = g.pool[0];
                // PDF::sample (PDF.h:124-129)
                const int offset = (int)(roundf(m.i_roughness * (1024 - 1)) * 1024);
                int index = (int)roundf(factor * (1024 - 1));
                index = max(0, min(1023, index));
                const float c_theta = sc.pdf_lookup[offset + index];
                n_pdf++;
                const float azimuth = g.pool[1] * 2.0f * SRX_FPI;
                const float c_theta_sin = full_sin(c_theta + SRX_FPI);
                const float c_theta_cos = full_cos(c_theta + SRX_FPI);
                f3 MN_1(-c_theta_sin * full_cos(azimuth), -c_theta_sin * full_sin(azimuth), -c_theta_cos);
                MN_1 = apply_tbn(MN_1, T, BT, normal);
                const f3 MN_2 = reflect3(MN_1, normal);
                const float cd2 = cdot3(flipped_output, MN_2);
                const float choice_prob = cd2 / (cdot3(flipped_output, MN_1) + cd2);
                f3 MN = MN_1;
                if (g.pool[2] < choice_prob) MN = MN_2;
                const float ref_prob = m.fast_fresnel_s(MN, flipped_output);
                f3 slope;
                f3 rc = throughput;
                bool is_transmissive = false;
                if (g.pool[3] < ref_prob) {
                    slope = reflect3(flipped_output, MN);
                    m.set_input(slope, sc.arccos_lookup);
                    rc = rc * m.PDF_specular_BRDF();
                } else if (g.pool[4] < m.transmission) {
                    const float dot_HI = dot3(MN, flipped_output);
                    const float f1 = sqrtf(1 - ior_u * ior_u * (1 - dot_HI * dot_HI));
                    slope = -(f1 * MN + ior_u * (flipped_output - dot_HI * MN));
                    is_transmissive = true;
                } else {
                    const float r1 = g.fRand(0, 1);                    // VecLib.cpp:372-382
                    const float r2 = g.fRand(0, 1);
                    const float r = sqrtf(r1);
                    const float theta = (float)(2 * SRX_PI * r2);
                    const float hx = r * full_cos(theta);
                    const float hy = r * full_sin(theta);
                    const f3 hemi(hx, sqrtf(1 - r1), hy);
                    slope = apply_tbn(f3(hemi.x, hemi.z, hemi.y), T, BT, normal);
                    m.set_input(slope, sc.arccos_lookup);
                    rc = rc * m.oren_nayar();
                }
                const bool t_flip = is_transmissive;
                if (t_flip != (dot3(slope, normal) < 0)) {
                    throughput = rc;                                   // quirk Q8: carried into the retry
                } else if (t_flip != (dot3(slope, geo_normal) < 0)) {
                    throughput = f3(1, 1, 1);
                } else if (rc.x < 0 || rc.y < 0 || rc.z < 0) {
                    throughput = f3(1, 1, 1);
                } else {
                    const float russian_p = std_max(rc.x, std_max(rc.y, rc.z));
                    if (g.pool[5] > russian_p) {
                        throughput = f3(1, 1, 1);
                        i++;                                           // slot consumed, no child
                    } else {
                        rc = rc / russian_p;
                        const f3 cpos = position + (float)((is_transmissive ? -1 : 1) * 0.0001) * geo_normal;
                        const f3 coef = rc / (float)bounce_count;      // Core.h:594
                        const f3 cthr = thr * coef;
                        const uint64_t ck = g.child_key();
                        child.ox = cpos.x; child.oy = cpos.y; child.oz = cpos.z;
                        child.dx = slope.x; child.dy = slope.y; child.dz = slope.z;
                        child.cr = cthr.x; child.cg = cthr.y; child.cb = cthr.z;
                        child.pixel = ray.pixel;
                        child.key_lo = (uint32_t)ck; child.key_hi = (uint32_t)(ck >> 32);
                        emit = true;
                        throughput = f3(1, 1, 1);
                        i++;
                    }
                }
                if (i >= bounce_count) active = false;
            }
        }
        // compaction: survivors are staged in this wave's LDS buffer (ballot + prefix count give the slot);
        // the buffer is flushed to the queue with ONE reservation whenever a trip's 64 lanes might not fit any more
        const unsigned long long em = __ballot(emit);
        if (em) {
            if (emit) {
                stage[staged + (uint32_t)__popcll(em & lt_mask)] = child;
                n_children++;
            }
            staged += (uint32_t)__popcll(em);
            if (staged > (uint32_t)(SRX_STAGE - 64)) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.out_count, staged);
                base = (uint32_t)__shfl((int)base, 0, 64);
                for (uint32_t k = lane; k < staged; k += 64) a.out_rays[base + k] = stage[k];
                staged = 0;
            }
        }
        // rounds: once every lane has had the trips a ray with few rejections needs (spill_after), a wave that is down to a few
        // busy lanes parks them and leaves (the next round packs the parked loops of the whole launch 64 to a wave)
        trips++;
        if (a.cont_out && trips >= a.spill_after && __popcll(__ballot(active)) <= a.spill_below) break;
    }
    // what is left is flushed with one reservation per block, and so are the parked loops
    const unsigned long long parked_bal = __ballot(active);
    __syncthreads();
    if (lane == 0) { s_wave_count[wave] = staged; s_wave_parked[wave] = (uint32_t)__popcll(parked_bal); }
    __syncthreads();
    {
        uint32_t before = 0, total = 0, p_before = 0, p_total = 0;
#pragma unroll
        for (int wv = 0; wv < WAVES; wv++) {
            const uint32_t c = s_wave_count[wv], pc = s_wave_parked[wv];
            if (wv < wave) { before += c; p_before += pc; }
            total += c; p_total += pc;
        }
        if (threadIdx.x == 0 && total) s_base[1] = atomicAdd(a.out_count, total);
        if (threadIdx.x == 0 && p_total) s_base[2] = atomicAdd(a.cont_count, p_total);
        __syncthreads();
        if (total) {
            const uint32_t base = s_base[1] + before;
            for (uint32_t k = lane; k < staged; k += 64) a.out_rays[base + k] = stage[k];
        }
        if (active) {                                 // only with a.cont_out: without it the loop runs until no lane is active
            const uint32_t at = s_base[2] + p_before + (uint32_t)__popcll(parked_bal & lt_mask);
            if (at < a.cont_capacity) {
                GCont c;
                c.idx = slot; c.i = (uint32_t)i; c.ri = (uint32_t)ri; c.tr = throughput.x; c.tg = throughput.y; c.tb = throughput.z; c.s = g.s;
                a.cont_out[at] = c;
            }
        }
    }

    // ---- counters: one atomic per wave and counter
    const uint32_t w_hits = resumed ? 0u : (uint32_t)__popcll(__ballot(is_hit));
    const uint32_t w_tex = wave_sum_u32(n_texels), w_pdf = wave_sum_u32(n_pdf), w_child = wave_sum_u32(n_children);
    if (lane == 0 && a.counters) {
        unsigned long long* c = a.counters + (size_t)(blockIdx.x & a.counter_slot_mask) * a.counter_slot_stride;
        if (w_hits) atomicAdd(&c[4], (unsigned long long)w_hits);
        if (w_tex) atomicAdd(&c[5], (unsigned long long)w_tex);
        if (w_pdf) atomicAdd(&c[6], (unsigned long long)w_pdf);
        if (w_child) atomicAdd(&c[7], (unsigned long long)w_child);
    }
}

// ---------------------------------------------------------------- primary rays (Camera.h:86-97,134-140)
template <class Gen>
__device__ __forceinline__ f3 camera_slope(const GCamera& c, int px, int py, uint32_t pass_index, Gen& g) {
    const float partial_width = c.lens_width / c.res_x;                // RectLens::_prep_function :50-85
    const float partial_height = c.lens_height / c.res_y;
    const float offset_x = -c.lens_width / 2;
    const float offset_y = -c.lens_height / 2;
    const float subdiv_w = partial_width / c.subdiv;
    const float subdiv_h = partial_height / c.subdiv;
    const int sub = (int)(pass_index % (uint32_t)(c.subdiv * c.subdiv));
    const int sx = sub % c.subdiv, sy = sub / c.subdiv;
    const float so_x = (float)(subdiv_w * (sx + 0.5));
    const float so_y = (float)(subdiv_h * (sy + 0.5));
    const float out_x = offset_x + partial_width * px;
    const float out_y = offset_y + partial_height * py;
    const float sdo_x = so_x - subdiv_w * 0.5f, sdo_y = so_y - subdiv_h * 0.5f;
    const float jx = g.fRand(0, subdiv_w);
    const float jy = g.fRand(0, subdiv_h);
    const float pos_x = (out_x + sdo_x) + jx, pos_y = (out_y + sdo_y) + jy;
    const f3 lens = f3(pos_x, 0.0f, pos_y) + f3(c.focal[0], c.focal[1], c.focal[2]);
    const float f1 = 0.0f - lens.x, f2 = 0.0f - lens.y, f3_ = 0.0f - lens.z;   // XYZ::distance, XYZ.cpp:259
    const float dist = sqrtf(f1 * f1 + f2 * f2 + f3_ * f3_);
    const f3 slope = (lens - f3(0.0f)) / dist;
    return apply_rotation(slope, c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
}

// slot_base: added to the pixel index a ray carries (the slot of its sample sums) when several passes are in flight
__global__ __launch_bounds__(256) void k_raygen(GCamera c, int x0, int w, GRows rm, int j0, int rows, uint32_t pass_index, uint64_t seed,
                                               uint32_t slot_base, GRay* rays, float* dir3, unsigned long long* keys) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)(w * rows)) return;
    const int px = x0 + (int)(idx % (uint32_t)w), py = srx_row_of(rm, j0 + (int)(idx / (uint32_t)w));
    const uint64_t npix = (uint64_t)c.res_x * c.res_y;
    const uint32_t pixel = (uint32_t)py * c.res_x + px;
    KeyedGen g;
    g.s = KeyedGen::primary_key(seed, (uint64_t)pass_index * npix + pixel);
    const f3 d = camera_slope(c, px, py, pass_index, g);
    if (rays) {
        GRay r;
        r.ox = c.pos[0]; r.oy = c.pos[1]; r.oz = c.pos[2];
        r.dx = d.x; r.dy = d.y; r.dz = d.z;
        r.cr = 1.0f; r.cg = 1.0f; r.cb = 1.0f;
        r.pixel = slot_base + pixel;
        r.key_lo = (uint32_t)g.s; r.key_hi = (uint32_t)(g.s >> 32);
        rays[idx] = r;
    }
    if (dir3) { dir3[3 * (size_t)idx] = d.x; dir3[3 * (size_t)idx + 1] = d.y; dir3[3 * (size_t)idx + 2] = d.z; }
    if (keys) keys[idx] = g.s;
}

// ---- primary rays and raw streams from the reference's own generator (parity ray sets of fixed XorRandom seeds)
// One sequential stream for the whole rectangle, two draws per pixel in row-major order: the thread of row r skips the
// 2 w r draws of the rows above it first (a sequential generator; the sets are 65 536 rays, not frames).
__global__ __launch_bounds__(64) void k_raygen_xorgen(GCamera c, int x0, int y0, int w, int h, uint32_t pass_index, uint64_t s0, uint64_t s1,
                                                     uint64_t s2, uint64_t s3, float* dir3) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= h) return;
    XorGen g;
    g.seed(s0, s1, s2, s3);
    for (long long k = 0; k < 2ll * w * row; k++) (void)g.xe_next();
    for (int x = 0; x < w; x++) {
        const f3 d = camera_slope(c, x0 + x, y0 + row, pass_index, g);
        float* o = dir3 + 3 * ((size_t)row * w + x);
        o[0] = d.x; o[1] = d.y; o[2] = d.z;
    }
}
__global__ void k_xorgen_stream(uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3, uint32_t n_pools, uint32_t n_scalar, float* pools8, float* scalars) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    XorGen g;
    g.seed(s0, s1, s2, s3);
    for (uint32_t i = 0; i < n_pools; i++) { g.regen_pool(); for (int k = 0; k < 8; k++) pools8[8 * (size_t)i + k] = g.pool[k]; }
    for (uint32_t i = 0; i < n_scalar; i++) scalars[i] = g.fRand(0.0f, 1.0f);
}
extern "C" hipError_t srx_launch_raygen_xorgen(const GCamera& c, int x0, int y0, int w, int h, uint32_t pass_index, const uint64_t* seed4, float* dir3,
                                               hipStream_t stream) {
    hipLaunchKernelGGL(k_raygen_xorgen, dim3((h + 63) / 64), dim3(64), 0, stream, c, x0, y0, w, h, pass_index, seed4[0], seed4[1], seed4[2], seed4[3], dir3);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_xorgen_stream(const uint64_t* seed4, uint32_t n_pools, uint32_t n_scalar, float* pools8, float* scalars, hipStream_t stream) {
    hipLaunchKernelGGL(k_xorgen_stream, dim3(1), dim3(64), 0, stream, seed4[0], seed4[1], seed4[2], seed4[3], n_pools, n_scalar, pools8, scalars);
    return hipGetLastError();
}

// ---- adaptive visit (RenderThread::process_iteratively, SceneManager.h:193-237)
// does block `block` take a sample on this visit? floor(p) + [u < frac(p)] > 0 with u keyed by (seed, visit, block)
__device__ __forceinline__ bool block_fires(const AdaptiveArgs& a, uint32_t block) {
    const float scaled = a.block_priority ? a.block_priority[block] : 1.0f;
    KeyedGen g;
    g.s = KeyedGen::mix(a.seed ^ KeyedGen::mix((uint64_t)a.visit_index * a.n_blocks + block + 0x2545F4914F6CDD1Dull) ^
                        KeyedGen::mix(0x51ED270B0BADC0DEull + 0x9E6C63D0676A9A99ull));
    const float u = g.fRand(0, 1);
    const float fl = floorf(scaled);
    const bool add_one = u < (scaled - fl);
    return (int)((float)add_one + fl) > 0;
}

__global__ __launch_bounds__(256) void k_raygen_adaptive(GCamera c, int x0, int w, GRows rm, int j0, int rows, AdaptiveArgs a, GRay* rays,
                                                        uint32_t* count) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t s_wave_count[4];
    __shared__ uint32_t s_base;
    bool fire = false;
    int px = 0, py = 0;
    uint32_t pixel = 0, n = 0;
    if (idx < (uint32_t)(w * rows)) {
        px = x0 + (int)(idx % (uint32_t)w); py = srx_row_of(rm, j0 + (int)(idx / (uint32_t)w));
        pixel = (uint32_t)py * c.res_x + px;
        fire = block_fires(a, (uint32_t)(py / 4) * a.blocks_x + (uint32_t)(px / 4));
        if (fire) n = (uint32_t)(int)a.accum4[pixel].w;
    }
    // compaction: one queue reservation per workgroup
    const unsigned long long bal = __ballot(fire);
    if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int wv = 0; wv < 4; wv++) { const uint32_t cnt = s_wave_count[wv]; if (wv < wave) before += cnt; total += cnt; }
    if (threadIdx.x == 0 && total) s_base = atomicAdd(count, total);
    __syncthreads();
    if (!fire) return;
    const uint32_t slot = s_base + before + (uint32_t)__popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    const uint64_t npix = (uint64_t)c.res_x * c.res_y;
    KeyedGen g;
    // stratum from the pixel's own count (iterations_done % samples_per_pixel, :212); the stream is keyed by the visit
    // so that a rejected (non-finite) sample is followed by a different draw, like the reference's redraw (:220-223)
    g.s = KeyedGen::primary_key(a.seed, (uint64_t)a.visit_index * npix + pixel);
    const f3 d = camera_slope(c, px, py, n, g);
    GRay r;
    r.ox = c.pos[0]; r.oy = c.pos[1]; r.oz = c.pos[2];
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.cr = 1.0f; r.cg = 1.0f; r.cb = 1.0f;
    r.pixel = pixel;
    r.key_lo = (uint32_t)g.s; r.key_hi = (uint32_t)(g.s >> 32);
    rays[slot] = r;
}

__global__ __launch_bounds__(256) void k_accumulate_adaptive(int res_x, int x0, int w, GRows rm, int rows, AdaptiveArgs a, float* sample3,
                                                            float4* accum4, float4* aux4) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)(w * rows)) return;
    const int px = x0 + (int)(idx % (uint32_t)w), py = srx_row_of(rm, (int)(idx / (uint32_t)w));
    if (!block_fires(a, (uint32_t)(py / 4) * a.blocks_x + (uint32_t)(px / 4))) return;
    const size_t pixel = (size_t)py * res_x + px;
    const f3 res(sample3[3 * pixel], sample3[3 * pixel + 1], sample3[3 * pixel + 2]);
    sample3[3 * pixel] = 0.0f; sample3[3 * pixel + 1] = 0.0f; sample3[3 * pixel + 2] = 0.0f;
    if (!(well_defined(res.x) && well_defined(res.y) && well_defined(res.z))) return;
    float4 o = accum4[pixel];
    const int n = (int)o.w;
    f3 cur(o.x, o.y, o.z);
    cur = cur * (float)((double)n / (n + 1));
    cur = cur + (float)(1.0 / (n + 1)) * res;
    accum4[pixel] = make_float4(cur.x, cur.y, cur.z, (float)(n + 1));
    if (aux4 && (n & 1)) {                                            // add_aux toggles per block iteration (:226-233)
        float4 x = aux4[pixel];
        const int na = (int)x.w;
        f3 ca(x.x, x.y, x.z);
        ca = ca * (float)((double)na / (na + 1));
        ca = ca + (float)(1.0 / (na + 1)) * res;
        aux4[pixel] = make_float4(ca.x, ca.y, ca.z, (float)(na + 1));
    }
}

// explicit ray batch -> queue records (srx_process_rays): pixel = ray index, throughput 1
__global__ __launch_bounds__(256) void k_pack_rays(const float* org3, const float* dir3, const unsigned long long* keys, uint32_t n, GRay* rays) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    GRay r;
    r.ox = org3[3 * (size_t)idx]; r.oy = org3[3 * (size_t)idx + 1]; r.oz = org3[3 * (size_t)idx + 2];
    r.dx = dir3[3 * (size_t)idx]; r.dy = dir3[3 * (size_t)idx + 1]; r.dz = dir3[3 * (size_t)idx + 2];
    r.cr = r.cg = r.cb = 1.0f;
    r.pixel = idx;
    r.key_lo = (uint32_t)keys[idx]; r.key_hi = (uint32_t)(keys[idx] >> 32);
    rays[idx] = r;
}

// ---------------------------------------------------------------- accumulation (SceneManager.h:220-233)
// accum4/aux4: float4 per pixel = (running mean rgb, sample count). A non-finite sample is skipped
// (the reference redraws it); the aux buffer takes the odd passes (add_aux toggles per pass).
// n_passes samples per pixel (passes pass_index .. pass_index + n_passes - 1, sums at slot p * slot_stride + pixel) are
// folded in one after the other, exactly as n_passes calls with one pass each would
__global__ __launch_bounds__(256) void k_accumulate(int res_x, int x0, int w, GRows rm, int rows, uint32_t pass_index0, int n_passes,
                                                   uint32_t slot_stride, float* sample3, float4* accum4, float4* aux4) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)(w * rows)) return;
    const int px = x0 + (int)(idx % (uint32_t)w), py = srx_row_of(rm, (int)(idx / (uint32_t)w));
    const size_t pixel = (size_t)py * res_x + px;
  for (int p = 0; p < n_passes; p++) {
    const uint32_t pass_index = pass_index0 + (uint32_t)p;
    const size_t slot = (size_t)p * slot_stride + pixel;
    const f3 res(sample3[3 * slot], sample3[3 * slot + 1], sample3[3 * slot + 2]);
    sample3[3 * slot] = 0.0f; sample3[3 * slot + 1] = 0.0f; sample3[3 * slot + 2] = 0.0f;
    if (!(well_defined(res.x) && well_defined(res.y) && well_defined(res.z))) continue;
    {
        float4 o = accum4[pixel];
        const int n = (int)o.w;
        f3 cur(o.x, o.y, o.z);
        cur = cur * (float)((double)n / (n + 1));
        cur = cur + (float)(1.0 / (n + 1)) * res;
        accum4[pixel] = make_float4(cur.x, cur.y, cur.z, (float)(n + 1));
    }
    if (aux4 && (pass_index & 1u)) {
        float4 o = aux4[pixel];
        const int n = (int)o.w;
        f3 cur(o.x, o.y, o.z);
        cur = cur * (float)((double)n / (n + 1));
        cur = cur + (float)(1.0 / (n + 1)) * res;
        aux4[pixel] = make_float4(cur.x, cur.y, cur.z, (float)(n + 1));
    }
  }
}

// ---------------------------------------------------------------- the rays of a batch that hit something
__global__ __launch_bounds__(256) void k_hitlist(const GHit* hits, uint32_t n, const uint32_t* n_ptr, uint32_t* list, uint32_t* count) {
    __shared__ uint32_t s_cnt[4][4];
    __shared__ uint32_t s_base;
    const uint32_t n_live = n_ptr ? min(*n_ptr, n) : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t first = blockIdx.x * 1024u;
    if (first >= n_live) return;
    bool h[4];
    unsigned long long bal[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t r = first + (uint32_t)t * 256u + threadIdx.x;
        h[t] = r < n_live && hits[r].material != 0xFFFFFFFFu;
        bal[t] = __ballot(h[t]);
        if (lane == 0) s_cnt[t][wave] = (uint32_t)__popcll(bal[t]);
    }
    __syncthreads();
    uint32_t before[4], total = 0;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int wv = 0; wv < 4; wv++) {
            if (wv == wave) before[t] = total;
            total += s_cnt[t][wv];
        }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(count, total) : 0u;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (h[t]) list[s_base + before[t] + (uint32_t)__popcll(bal[t] & lt_mask)] = first + (uint32_t)t * 256u + threadIdx.x;
}

// ---------------------------------------------------------------- launchers
extern "C" hipError_t srx_launch_hitlist(const GHit* hits, uint32_t n, const uint32_t* n_ptr, uint32_t* list, uint32_t* count, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hitlist, dim3((n + 1023u) / 1024u), dim3(256), 0, stream, hits, n, n_ptr, list, count);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_shade(const GScene& sc, const ShadeArgs& a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    // waves per block by the generation's fan-out (experiments: SRX_SHADE_WAVES_BY_FANOUT="w32,w8,w2,w1")
    static int by_fanout[4] = {SRX_SHADE_WAVES, SRX_SHADE_WAVES, SRX_SHADE_WAVES, SRX_SHADE_WAVES};
    static const bool parsed = [] {
        if (const char* e = getenv("SRX_SHADE_WAVES_BY_FANOUT")) (void)sscanf(e, "%d%*c%d%*c%d%*c%d", &by_fanout[0], &by_fanout[1], &by_fanout[2], &by_fanout[3]);
        return true;
    }();
    (void)parsed;
    const int fan = sc.bounce_count[a.generation < 32 ? a.generation : 31];
    const int waves = by_fanout[fan >= 32 ? 0 : (fan >= 8 ? 1 : (fan >= 2 ? 2 : 3))];
    const unsigned threads = 64u * (unsigned)(waves == 1 ? 1 : (waves == 2 ? 2 : 4));
    const dim3 grid((a.n + threads - 1) / threads);
    if (threads == 64) hipLaunchKernelGGL(k_shade<1>, grid, dim3(threads), 0, stream, sc, a);
    else if (threads == 128) hipLaunchKernelGGL(k_shade<2>, grid, dim3(threads), 0, stream, sc, a);
    else hipLaunchKernelGGL(k_shade<4>, grid, dim3(threads), 0, stream, sc, a);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_raygen(const GCamera& c, int x0, int w, GRows rows_map, int j0, int rows, uint32_t pass_index,
                                        uint64_t seed, uint32_t slot_base, GRay* rays, float* dir3, unsigned long long* keys,
                                        hipStream_t stream) {
    const uint32_t n = (uint32_t)(w * rows);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, stream, c, x0, w, rows_map, j0, rows, pass_index, seed, slot_base, rays,
                       dir3, keys);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_raygen_adaptive(const GCamera& c, int x0, int w, GRows rows_map, int j0, int rows, AdaptiveArgs a,
                                                 GRay* rays, uint32_t* count, hipStream_t stream) {
    const uint32_t n = (uint32_t)(w * rows);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raygen_adaptive, dim3((n + 255) / 256), dim3(256), 0, stream, c, x0, w, rows_map, j0, rows, a, rays, count);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_accumulate_adaptive(int res_x, int x0, int w, GRows rows_map, int rows, AdaptiveArgs a, float* sample3,
                                                     float* accum4, float* aux4, hipStream_t stream) {
    const uint32_t n = (uint32_t)(w * rows);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_accumulate_adaptive, dim3((n + 255) / 256), dim3(256), 0, stream, res_x, x0, w, rows_map, rows, a, sample3,
                       (float4*)accum4, (float4*)aux4);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_pack_rays(const float* org3, const float* dir3, const unsigned long long* keys, uint32_t n,
                                           GRay* rays, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_rays, dim3((n + 255) / 256), dim3(256), 0, stream, org3, dir3, keys, n, rays);
    return hipGetLastError();
}
extern "C" hipError_t srx_launch_accumulate(int res_x, int x0, int w, GRows rows_map, int rows, uint32_t pass_index, int n_passes,
                                            uint32_t slot_stride, float* sample3, float* accum4, float* aux4, hipStream_t stream) {
    const uint32_t n = (uint32_t)(w * rows);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_accumulate, dim3((n + 255) / 256), dim3(256), 0, stream, res_x, x0, w, rows_map, rows, pass_index, n_passes,
                       slot_stride, sample3, (float4*)accum4, (float4*)aux4);
    return hipGetLastError();
}
