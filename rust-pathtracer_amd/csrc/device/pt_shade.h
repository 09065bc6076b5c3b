// pt_shade.h — device functions of the integrator: everything `trace_inner`
// (renderer/pathtracer.rs:101-326) calls between two `scene.hit` queries.
// Arithmetic order follows the cited reference lines exactly; control flow is
// re-cut into a per-bounce state machine by pt_core.hip (wavefront design).
#pragma once
#include "pt_detmath.h"
#include "pt_types.h"

namespace ptd {

// ------------------------------------------------------------------ Rng -------
// PCG-XSH-RR 64/32 (renderer/subsurface.rs:15-35).  next() divides by
// `u32::MAX as f32`, which rounds to 2^32, so the result lies in [0, 1].
struct Rng {
    uint64_t state;
    PT_DEV uint32_t next_u32() {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        const uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    PT_DEV float next() { return (float)next_u32() / 4294967296.0f; }
};
PT_DEV Rng rng_new(uint32_t seed) { Rng r; r.state = (uint64_t)seed; r.next_u32(); return r; }

// ------------------------------------------------------------- textures -------
// Texture::sample (texture.rs:64-101): wrap-repeat bilinear over f32 RGB texels.
PT_DEV V3 texture_sample(const TexRec& t, const float* texels, float uvx, float uvy, uint32_t& lookups) {
    if (t.offset == ~0ull) return v3(t.solid[0], t.solid[1], t.solid[2]);
    lookups++;
    const float u = rem_euclid1(uvx), v = rem_euclid1(uvy);
    const float px = u * (float)t.width - 0.5f, py = v * (float)t.height - 0.5f;
    const float fpx = floorf(px), fpy = floorf(py);
    const int x0 = (int)fpx, y0 = (int)fpy;
    const int w = (int)t.width, h = (int)t.height;
    const uint32_t x0w = (uint32_t)rem_euclid_i(x0, w), y0h = (uint32_t)rem_euclid_i(y0, h);
    const uint32_t x1w = (uint32_t)rem_euclid_i(x0 + 1, w), y1h = (uint32_t)rem_euclid_i(y0 + 1, h);
    const float fx = px - fpx, fy = py - fpy;
    const float* base = texels + t.offset;
    const float* p00 = base + ((size_t)y0h * t.width + x0w) * 3;
    const float* p10 = base + ((size_t)y0h * t.width + x1w) * 3;
    const float* p01 = base + ((size_t)y1h * t.width + x0w) * 3;
    const float* p11 = base + ((size_t)y1h * t.width + x1w) * 3;
    const V3 c00 = v3(p00[0], p00[1], p00[2]), c10 = v3(p10[0], p10[1], p10[2]);
    const V3 c01 = v3(p01[0], p01[1], p01[2]), c11 = v3(p11[0], p11[1], p11[2]);
    const V3 c0 = lerp(c00, c10, fx);
    const V3 c1 = lerp(c01, c11, fx);
    return lerp(c0, c1, fy);
}

// ------------------------------------------------------------------ sky -------
// TextureSky runtime (sky/texture.rs:41-65, 114-211)
struct ShadeCnt { uint32_t hitdata, material, texture, cdf_probes, sky_pdf, sky_texel, sss_steps; };

PT_DEV V3 sky_rotate(const SceneDev& S, V3 d) {
    if (S.sky_rotation == 0.0f) return d;
    float s, c;
    d_sincosf(S.sky_rotation, &s, &c);
    return v3(d.x * c - d.z * s, d.y, d.x * s + d.z * c);
}
PT_DEV V3 sky_unrotate(const SceneDev& S, V3 d) {
    if (S.sky_rotation == 0.0f) return d;
    float s, c;
    d_sincosf(S.sky_rotation, &s, &c);
    return v3(d.x * c + d.z * s, d.y, -d.x * s + d.z * c);
}
PT_DEV void direction_to_uv(V3 d, float* u, float* v) {
    *u = 0.5f + d_atan2f(d.z, d.x) / (2.0f * PT_PI);
    *v = 0.5f - d_asinf(d.y) / PT_PI;
}
PT_DEV V3 sky_sample(const SceneDev& S, V3 direction, ShadeCnt& sc) {
    float u, v;
    direction_to_uv(sky_rotate(S, direction), &u, &v);
    return texture_sample(S.sky_tex, S.texels, u, v, sc.sky_texel);
}
PT_DEV uint32_t f2u_sat(float f) {   // Rust `as u32`
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
PT_DEV float sky_pdf(const SceneDev& S, V3 direction, ShadeCnt& sc) {
    if (S.sky_total_weight <= 0.0f) return 1.0f / (4.0f * PT_PI);
    sc.sky_pdf++;
    float u, v;
    direction_to_uv(sky_rotate(S, direction), &u, &v);
    const uint32_t W = S.sky_w, H = S.sky_h;
    uint32_t x = f2u_sat(u * (float)W); x = x < W - 1 ? x : W - 1;
    uint32_t y = f2u_sat(v * (float)H); y = y < H - 1 ? y : H - 1;
    const float marginal_pdf = S.sky_marginal[y + 1] - S.sky_marginal[y];
    const size_t row = (size_t)y * (W + 1);
    const float conditional_pdf = S.sky_conditional[row + x + 1] - S.sky_conditional[row + x];
    const float pdf_uv = marginal_pdf * conditional_pdf * (float)(W * H);
    const float sin_theta = f_max(d_sinf(PT_PI * v), 1e-10f);
    return pdf_uv / (2.0f * PT_PI * PT_PI * sin_theta);
}
// upper_bound - 1 (sky/texture.rs:127-139)
PT_DEV uint32_t cdf_search(const float* cdf, uint32_t len, float u, uint32_t& probes) {
    uint32_t lo = 0, hi = len - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        probes++;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo == 0 ? 0 : lo - 1;
}
// The same index without the dependent chain of ~log2(len) probes: for a non-decreasing CDF the loop above returns
// lo = #{i <= len-2 : cdf[i] <= u}, and that count is found 16 entries (one 64-B sector) per level.
PT_DEV uint32_t cdf_count16(const float* tab, const uint32_t* off, const uint32_t* stride, uint32_t row, uint32_t levels, float u) {
    uint32_t g = 0;
    for (int l = (int)levels - 1; l >= 0; --l) {
        const float4* p = reinterpret_cast<const float4*>(tab + off[l] + (size_t)row * stride[l] + 16u * g);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        const uint32_t n = (uint32_t)(a.x <= u) + (uint32_t)(a.y <= u) + (uint32_t)(a.z <= u) + (uint32_t)(a.w <= u) +
                           (uint32_t)(b.x <= u) + (uint32_t)(b.y <= u) + (uint32_t)(b.z <= u) + (uint32_t)(b.w <= u) +
                           (uint32_t)(c.x <= u) + (uint32_t)(c.y <= u) + (uint32_t)(c.z <= u) + (uint32_t)(c.w <= u) +
                           (uint32_t)(d.x <= u) + (uint32_t)(d.y <= u) + (uint32_t)(d.z <= u) + (uint32_t)(d.w <= u);
        g = 16u * g + n;
    }
    return g;
}
struct SkySample { V3 direction, radiance; float pdf; };
// sample_direction (sky/texture.rs:149-188) in two halves: the CDF searches (u1, u2) -> (u, v), and everything behind them.
struct SkyUV { float u, v; };
// FAST: use the 16-ary tables when the scene has them (the counting kernels keep the reference's probe sequence)
template <bool FAST>
PT_DEV SkyUV sky_search_uv(const SceneDev& S, float u1, float u2, ShadeCnt& sc) {
    const uint32_t W = S.sky_w, H = S.sky_h;
    const uint32_t zero_stride[4] = {0u, 0u, 0u, 0u};
    uint32_t y;
    if (FAST && S.cdf_levels_m) { const uint32_t lo = cdf_count16(S.sky_search, S.cdf_off_m, zero_stride, 0u, S.cdf_levels_m, u1); y = lo == 0u ? 0u : lo - 1u; }
    else y = cdf_search(S.sky_marginal, H + 1, u1, sc.cdf_probes);
    const float v_lo = S.sky_marginal[y], v_hi = S.sky_marginal[y + 1];
    const float dv = (u1 - v_lo) / (v_hi - v_lo + 1e-10f);
    SkyUV r;
    r.v = ((float)y + dv) / (float)H;
    const float* row_cdf = S.sky_conditional + (size_t)y * (W + 1);
    uint32_t x;
    if (FAST && S.cdf_levels_c) { const uint32_t lo = cdf_count16(S.sky_search, S.cdf_off_c, S.cdf_stride_c, y, S.cdf_levels_c, u2); x = lo == 0u ? 0u : lo - 1u; }
    else x = cdf_search(row_cdf, W + 1, u2, sc.cdf_probes);
    const float u_lo = row_cdf[x], u_hi = row_cdf[x + 1];
    const float du = (u2 - u_lo) / (u_hi - u_lo + 1e-10f);
    r.u = ((float)x + du) / (float)W;
    return r;
}
// The same searches as straight-line code for scenes whose tables have exactly three levels each (the upload pads
// shallower ones): no loop, no branch, so that the compiler can run this chain of six dependent reads BESIDE another
// chain of the same basic block (k_shade starts it ahead of the hit-data and material gathers: "Sky search ahead").
PT_DEV void load16f(const float* q, float4& a, float4& b, float4& c, float4& d) {
    const float4* p = reinterpret_cast<const float4*>(q);
    a = p[0]; b = p[1]; c = p[2]; d = p[3];
}
PT_DEV uint32_t count16f(const float4 a, const float4 b, const float4 c, const float4 d, float u) {
    return (uint32_t)(a.x <= u) + (uint32_t)(a.y <= u) + (uint32_t)(a.z <= u) + (uint32_t)(a.w <= u) +
           (uint32_t)(b.x <= u) + (uint32_t)(b.y <= u) + (uint32_t)(b.z <= u) + (uint32_t)(b.w <= u) +
           (uint32_t)(c.x <= u) + (uint32_t)(c.y <= u) + (uint32_t)(c.z <= u) + (uint32_t)(c.w <= u) +
           (uint32_t)(d.x <= u) + (uint32_t)(d.y <= u) + (uint32_t)(d.z <= u) + (uint32_t)(d.w <= u);
}
// compute_hit_data's two gathers (InstShade -> ShadeTri) and the sky searches' reads, issued as ONE interleaved sequence:
// a load is written where it can be issued, its use where the other chain's next load has been issued, so that each wait
// of one chain has a read of the other in flight beside it (loads return in order; the waits the compiler inserts count
// the younger ones that may stay outstanding).
PT_DEV SkyUV gather_hit_and_sky(const SceneDev& S, uint32_t inst, const uint32_t* hit_tri_slot, float u1, float u2,
                                float4& i0, float4& i1, float4& i2, float4& qa, float4& qb, float4& qc, float4& qd, float4& qe) {
    const uint32_t W = S.sky_w, H = S.sky_h;
    const float* t = S.sky_search;
    float4 a, b, c, d;
    const float4* isp = reinterpret_cast<const float4*>(&S.inst_shade[inst]);
    i0 = isp[0]; i1 = isp[1]; i2 = isp[2];                                   // hit 1
    const uint32_t tri_idx = *hit_tri_slot;
    load16f(t + S.cdf_off_m[2], a, b, c, d);                                 // sky 1
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the hit chain's loads to their first use, behind the whole search)
    uint32_t g = count16f(a, b, c, d, u1);
    load16f(t + S.cdf_off_m[1] + 16u * g, a, b, c, d);                       // sky 2
    const float4* q = reinterpret_cast<const float4*>(&S.shade_tris[(size_t)__float_as_uint(i2.z) + tri_idx]);
    qa = q[0]; qb = q[1]; qc = q[2]; qd = q[3]; qe = q[4];                   // hit 2
    __builtin_amdgcn_sched_barrier(0);
    g = 16u * g + count16f(a, b, c, d, u1);
    load16f(t + S.cdf_off_m[0] + 16u * g, a, b, c, d);                       // sky 3
    g = 16u * g + count16f(a, b, c, d, u1);
    const uint32_t y = g == 0u ? 0u : g - 1u;
    const float v_lo = S.sky_marginal[y], v_hi = S.sky_marginal[y + 1];      // sky 4
    load16f(t + S.cdf_off_c[2] + (size_t)y * S.cdf_stride_c[2], a, b, c, d);
    const float dv = (u1 - v_lo) / (v_hi - v_lo + 1e-10f);
    SkyUV r;
    r.v = ((float)y + dv) / (float)H;
    const float* row_cdf = S.sky_conditional + (size_t)y * (W + 1);
    uint32_t h = count16f(a, b, c, d, u2);
    load16f(t + S.cdf_off_c[1] + (size_t)y * S.cdf_stride_c[1] + 16u * h, a, b, c, d);     // sky 5
    h = 16u * h + count16f(a, b, c, d, u2);
    load16f(t + S.cdf_off_c[0] + (size_t)y * S.cdf_stride_c[0] + 16u * h, a, b, c, d);     // sky 6
    h = 16u * h + count16f(a, b, c, d, u2);
    const uint32_t x = h == 0u ? 0u : h - 1u;
    const float u_lo = row_cdf[x], u_hi = row_cdf[x + 1];                     // sky 7
    const float du = (u2 - u_lo) / (u_hi - u_lo + 1e-10f);
    r.u = ((float)x + du) / (float)W;
    return r;
}
PT_DEV SkySample sky_sample_at(const SceneDev& S, SkyUV uv, ShadeCnt& sc) {
    SkySample r;
    // uv_to_direction (sky/texture.rs:120-125)
    const float phi = (uv.u - 0.5f) * 2.0f * PT_PI;
    const float theta = (0.5f - uv.v) * PT_PI;
    float sph, cph, sth, cth;
    d_sincosf(phi, &sph, &cph);
    d_sincosf(theta, &sth, &cth);
    r.direction = sky_unrotate(S, v3(cth * sph, sth, cth * cph));
    r.radiance = texture_sample(S.sky_tex, S.texels, uv.u, uv.v, sc.sky_texel);
    r.pdf = sky_pdf(S, r.direction, sc);
    return r;
}
PT_DEV SkySample sky_sample_uniform(const SceneDev& S, float u1, float u2, ShadeCnt& sc) {
    SkySample r;
    const float phi = u1 * 2.0f * PT_PI;
    const float cos_theta = 1.0f - 2.0f * u2;
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float sp, cp;
    d_sincosf(phi, &sp, &cp);
    r.direction = v3(sin_theta * cp, cos_theta, sin_theta * sp);
    r.radiance = sky_sample(S, r.direction, sc);
    r.pdf = 1.0f / (4.0f * PT_PI);
    return r;
}
template <bool FAST>
PT_DEV SkySample sky_sample_direction(const SceneDev& S, float u1, float u2, ShadeCnt& sc) {
    if (S.sky_total_weight <= 0.0f) return sky_sample_uniform(S, u1, u2, sc);
    return sky_sample_at(S, sky_search_uv<FAST>(S, u1, u2, sc), sc);
}

// -------------------------------------------------------------- brdf.rs -------
PT_DEV float ggx_d(float n_dot_h, float alpha) {
    n_dot_h = f_max(n_dot_h, 0.0f);
    const float a2 = alpha * alpha;
    const float denom = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
    return a2 / (PT_PI * denom * denom);
}
PT_DEV float ggx_g1(float n_dot_v, float alpha) {
    const float a2 = alpha * alpha;
    const float denom = n_dot_v + sqrtf(a2 + (1.0f - a2) * n_dot_v * n_dot_v);
    return (2.0f * n_dot_v) / denom;
}
PT_DEV float ggx_g(float n_dot_l, float n_dot_v, float alpha) { return ggx_g1(n_dot_l, alpha) * ggx_g1(n_dot_v, alpha); }
PT_DEV float pdf_ggx(float n_dot_h, float v_dot_h, float alpha) {
    if (n_dot_h <= 0.0f || v_dot_h <= 0.0f) return 0.0f;
    const float d = ggx_d(n_dot_h, alpha);
    return d * n_dot_h / (4.0f * v_dot_h);
}
PT_DEV float pdf_cosine(float n_dot_l) { return n_dot_l / PT_PI; }
PT_DEV float pdf_brdf(float n_dot_l, float n_dot_h, float v_dot_h, float alpha, float spec_prob) {
    const float pdf_spec = pdf_ggx(n_dot_h, v_dot_h, alpha);
    const float pdf_diff = pdf_cosine(n_dot_l);
    return spec_prob * pdf_spec + (1.0f - spec_prob) * pdf_diff;
}
PT_DEV V3 fresnel_schlick(float cos_theta, V3 f0) {
    return f0 + (splat(1.0f) - f0) * powi5(f_clamp(1.0f - cos_theta, 0.0f, 1.0f));
}
PT_DEV float fresnel_dielectric(float cos_theta, float ior) {
    const float r0 = powi2((1.0f - ior) / (1.0f + ior));
    return r0 + (1.0f - r0) * powi5(f_clamp(1.0f - cos_theta, 0.0f, 1.0f));
}
PT_DEV bool refract(V3 incident, V3 normal, float eta, V3* out) {
    const float cos_i = -dot(incident, normal);
    const float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return false;
    const float cos_t = sqrtf(1.0f - sin2_t);
    *out = eta * incident + (eta * cos_i - cos_t) * normal;
    return true;
}
PT_DEV V3 reflect(V3 incident, V3 normal) { return normalize(incident - 2.0f * dot(incident, normal) * normal); }
PT_DEV float ggx_directional_albedo(float n_dot_v, float alpha) {
    return 1.0f - powi5(1.0f - n_dot_v) * powi2(1.0f - alpha) * 0.65f;
}
PT_DEV V3 multiscatter_compensation(float n_dot_v, float n_dot_l, float alpha, V3 f0) {
    const float e_v = ggx_directional_albedo(n_dot_v, alpha);
    const float e_l = ggx_directional_albedo(n_dot_l, alpha);
    const V3 one = splat(1.0f);
    const V3 f_avg = f0 + (one - f0) / 21.0f;
    const V3 f_ms = f_avg * f_avg * (1.0f - e_v) * (1.0f - e_l) / (PT_PI * (one - f_avg * (1.0f - e_v)));
    return one + f_ms / (e_v * e_l + 0.001f);
}
PT_DEV V3 evaluate_brdf(V3 v, V3 l, V3 n, float alpha, V3 f0, V3 diffuse_color) {
    const float n_dot_l = dot(n, l), n_dot_v = dot(n, v);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return splat(0.0f);
    const V3 h = normalize(v + l);
    const float n_dot_h = f_max(dot(n, h), 0.001f);
    const float v_dot_h = f_max(dot(v, h), 0.001f);
    const V3 f = fresnel_schlick(v_dot_h, f0);
    const float d = ggx_d(n_dot_h, alpha);
    const float g = ggx_g(n_dot_l, n_dot_v, alpha);
    const V3 ms = multiscatter_compensation(n_dot_v, n_dot_l, alpha, f0);
    const V3 specular = f * d * g / (4.0f * n_dot_l * n_dot_v) * ms;
    const V3 diffuse = diffuse_color * (splat(1.0f) - f) / PT_PI;
    return (specular + diffuse) * n_dot_l;
}

// ---------------------------------------------------------- sampling.rs -------
PT_DEV V3 sample_ggx(float u1, float u2, float alpha) {
    const float phi = 2.0f * PT_PI * u1;
    const float cos_theta = sqrtf((1.0f - u2) / (u2 * (alpha * alpha - 1.0f) + 1.0f));
    const float sin_theta = sqrtf(f_max(1.0f - cos_theta * cos_theta, 0.0f));
    float sp, cp;
    d_sincosf(phi, &sp, &cp);
    return v3(sin_theta * cp, sin_theta * sp, cos_theta);
}
PT_DEV V3 sample_cosine_hemisphere(float u1, float u2) {
    const float r = sqrtf(u1);
    const float phi = 2.0f * PT_PI * u2;
    float sp, cp;
    d_sincosf(phi, &sp, &cp);
    return v3(r * cp, r * sp, sqrtf(f_max(1.0f - u1, 0.0f)));
}
PT_DEV void build_basis(V3 n, V3* t, V3* b) {
    const V3 up = (fabsf(n.y) < 0.999f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    *t = normalize(cross(up, n));
    *b = cross(n, *t);
}
PT_DEV V3 to_world(V3 local, V3 t, V3 b, V3 n) { return t * local.x + b * local.y + n * local.z; }
PT_DEV void half_vector_dots(V3 v, V3 l, V3 n, float* n_dot_h, float* v_dot_h) {
    const V3 h = normalize(v + l);
    *n_dot_h = f_max(dot(n, h), 0.001f);
    *v_dot_h = f_max(dot(v, h), 0.001f);
}
PT_DEV float mis_weight(float pdf_a, float pdf_b) {
    const float a2 = pdf_a * pdf_a, b2 = pdf_b * pdf_b;
    return a2 / (a2 + b2 + 1e-10f);
}
PT_DEV float safe_inv_pdf(float pdf) { return f_min(1.0f / pdf, 100.0f); }
PT_DEV V3 clamp_radiance(V3 radiance) {
    if (!is_finite(radiance)) return splat(0.0f);
    const float max_component = max_element(radiance);
    if (max_component > 10.0f) return radiance * (10.0f / max_component);
    return radiance;
}
PT_DEV V3 fix_shading_normal(V3 shading_n, V3 geo_n, V3 v) {
    if (dot(shading_n, geo_n) < 0.0f) shading_n = -shading_n;
    const float n_dot_v = dot(shading_n, v);
    if (n_dot_v > 0.0f) return shading_n;
    const V3 projected = shading_n - v * n_dot_v;
    const float len = length(projected);
    if (len > 1e-6f) return normalize(projected / len * 0.5f + geo_n * 0.5f);
    return geo_n;
}

// ------------------------------------------------------------- hit data -------
// TriangleMesh::compute_hit_data (shape/triangle_mesh.rs:102-159) followed by
// the normal transform of Instance::hit (accel/mod.rs:134-135).  Done once per
// path vertex from the (instance, triangle, u, v) record the extend kernel wrote.
struct HitData {
    float t;
    uint32_t material_id;
    V3 normal, geo_normal, tangent, bitangent;
    float uvx, uvy;
};
PT_DEV V3 load3(const float* p, uint32_t base, uint32_t i) {
    const float* q = p + base + 3 * (size_t)i;
    return v3(q[0], q[1], q[2]);
}
// one ShadeTri from the mesh's vertex pools (upload-time kernel; see pt_types.h)
PT_DEV ShadeTri build_shade_tri(const SceneDev& S, const MeshRec& m, uint32_t tri_idx) {
    const uint32_t* tri = S.indices + m.idx_base + 3 * (size_t)tri_idx;
    const uint32_t i0 = tri[0], i1 = tri[1], i2 = tri[2];
    const V3 v0 = load3(S.positions, m.pos_base, i0), v1 = load3(S.positions, m.pos_base, i1), v2 = load3(S.positions, m.pos_base, i2);
    const V3 fn = normalize(cross(v1 - v0, v2 - v0));
    V3 n0 = splat(0.0f), n1 = n0, n2 = n0;
    if (m.nrm_base != 0xFFFFFFFFu) { n0 = load3(S.normals, m.nrm_base, i0); n1 = load3(S.normals, m.nrm_base, i1); n2 = load3(S.normals, m.nrm_base, i2); }
    float u0 = 0.0f, w0 = 0.0f, u1 = 0.0f, w1 = 0.0f, u2 = 0.0f, w2 = 0.0f;
    if (m.uv_base != 0xFFFFFFFFu) {
        const float* a = S.uvs + m.uv_base + 2 * (size_t)i0;
        const float* b = S.uvs + m.uv_base + 2 * (size_t)i1;
        const float* c = S.uvs + m.uv_base + 2 * (size_t)i2;
        u0 = a[0]; w0 = a[1]; u1 = b[0]; w1 = b[1]; u2 = c[0]; w2 = c[1];
    }
    ShadeTri r;
    r.a = make_float4(fn.x, fn.y, fn.z, n0.x);
    r.b = make_float4(n0.y, n0.z, n1.x, n1.y);
    r.c = make_float4(n1.z, n2.x, n2.y, n2.z);
    r.d = make_float4(u0, w0, u1, w1);
    r.e = make_float4(u2, w2, 0.0f, 0.0f);
    return r;
}

// the arithmetic half: InstShade as three float4 (n[0..8], mesh id, shade_base, mat_flags) and the ShadeTri already in registers
PT_DEV HitData hit_data_from(const float4 i0, const float4 i1, const float4 i2, const float4 qa, const float4 qb, const float4 qc,
                             const float4 qd, const float4 qe, float t, float u, float v) {
    const uint32_t mat_flags = __float_as_uint(i2.w);
    const float w = 1.0f - u - v;
    const V3 face_normal = v3(qa.x, qa.y, qa.z);
    V3 normal;
    if (mat_flags & SHADE_HAS_NORMALS) {
        const V3 n0 = v3(qa.w, qb.x, qb.y), n1 = v3(qb.z, qb.w, qc.x), n2 = v3(qc.y, qc.z, qc.w);
        normal = normalize(n0 * w + n1 * u + n2 * v);
    } else {
        normal = face_normal;
    }
    const V3 geo_normal = (dot(face_normal, normal) >= 0.0f) ? face_normal : -face_normal;
    HitData h;
    if (mat_flags & SHADE_HAS_UVS) {
        h.uvx = (qd.x * w + qd.z * u) + qe.x * v;
        h.uvy = (qd.y * w + qd.w * u) + qe.y * v;
    } else {
        h.uvx = u; h.uvy = v;
    }
    // tangent frame is built from the LOCAL normal and left untransformed (accel/mod.rs:134-135)
    const V3 up = (fabsf(normal.y) < 0.999f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    h.tangent = normalize(cross(up, normal));
    h.bitangent = cross(normal, h.tangent);
    h.t = t;
    h.material_id = mat_flags & SHADE_MATERIAL_MASK;
    // normal_matrix.transform_vector3(n).normalize()
    const float nm[9] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x};
    h.normal = normalize(v3((nm[0] * normal.x + nm[3] * normal.y) + nm[6] * normal.z,
                            (nm[1] * normal.x + nm[4] * normal.y) + nm[7] * normal.z,
                            (nm[2] * normal.x + nm[5] * normal.y) + nm[8] * normal.z));
    h.geo_normal = normalize(v3((nm[0] * geo_normal.x + nm[3] * geo_normal.y) + nm[6] * geo_normal.z,
                                (nm[1] * geo_normal.x + nm[4] * geo_normal.y) + nm[7] * geo_normal.z,
                                (nm[2] * geo_normal.x + nm[5] * geo_normal.y) + nm[8] * geo_normal.z));
    return h;
}
static_assert(sizeof(InstShade) == 48, "InstShade is read as three float4");
PT_DEV HitData compute_hit_data(const SceneDev& S, uint32_t inst, uint32_t tri_idx, float t, float u, float v) {
    const float4* isp = reinterpret_cast<const float4*>(&S.inst_shade[inst]);
    const float4 i0 = isp[0], i1 = isp[1], i2 = isp[2];
    const float4* q = reinterpret_cast<const float4*>(&S.shade_tris[(size_t)__float_as_uint(i2.z) + tri_idx]);
    const float4 qa = q[0], qb = q[1], qc = q[2], qd = q[3], qe = q[4];
    return hit_data_from(i0, i1, i2, qa, qb, qc, qd, qe, t, u, v);
}

// -------------------------------------------------------- sample_material -----
// renderer/pathtracer.rs:35-87
struct SampledMaterial { V3 base_color; float roughness, metallic; V3 emissive, normal; };
PT_DEV SampledMaterial sample_material(const SceneDev& S, const PtxMaterial& mat, const HitData& hit, ShadeCnt& sc) {
    SampledMaterial r;
    const V3 bcf = v3(mat.base_color_factor[0], mat.base_color_factor[1], mat.base_color_factor[2]);
    r.base_color = mat.base_color_texture >= 0
        ? texture_sample(S.textures[mat.base_color_texture], S.texels, hit.uvx, hit.uvy, sc.texture) * bcf : bcf;
    if (mat.metallic_roughness_texture >= 0) {
        const V3 s = texture_sample(S.textures[mat.metallic_roughness_texture], S.texels, hit.uvx, hit.uvy, sc.texture);
        r.roughness = s.y * mat.roughness_factor;
        r.metallic = s.z * mat.metallic_factor;
    } else {
        r.roughness = mat.roughness_factor;
        r.metallic = mat.metallic_factor;
    }
    const V3 ef = v3(mat.emissive_factor[0], mat.emissive_factor[1], mat.emissive_factor[2]);
    r.emissive = mat.emissive_texture >= 0
        ? texture_sample(S.textures[mat.emissive_texture], S.texels, hit.uvx, hit.uvy, sc.texture) * ef : ef;
    if (mat.normal_texture >= 0) {
        const V3 s = texture_sample(S.textures[mat.normal_texture], S.texels, hit.uvx, hit.uvy, sc.texture);
        V3 tn = normalize(s * 2.0f - splat(1.0f));
        tn = normalize(v3(tn.x * mat.normal_scale, tn.y * mat.normal_scale, tn.z));
        r.normal = normalize(hit.tangent * tn.x + hit.bitangent * tn.y + hit.normal * tn.z);
    } else {
        r.normal = hit.normal;
    }
    r.roughness = f_clamp(r.roughness, 0.04f, 1.0f);
    return r;
}

// renderer/pathtracer.rs:350-360
PT_DEV bool russian_roulette(Rng& rng, V3& throughput, uint32_t bounce) {
    if (bounce > 2u) {
        const float p = f_clamp(max_element(throughput), 0.05f, 0.95f);
        if (rng.next() > p) return false;
        throughput = throughput / p;
    }
    return true;
}
// renderer/pathtracer.rs:362-370
PT_DEV float compute_spec_prob(float n_dot_v, V3 f0, V3 diffuse_color) {
    const V3 spec_albedo = fresnel_schlick(n_dot_v, f0);
    const V3 diff_albedo = diffuse_color * (splat(1.0f) - spec_albedo);
    const V3 lum = v3(0.2126f, 0.7152f, 0.0722f);
    const float spec_lum = dot(spec_albedo, lum), diff_lum = dot(diff_albedo, lum);
    return f_clamp(spec_lum / (spec_lum + diff_lum + 0.001f), 0.1f, 0.9f);
}

// ---- ray-gen: BlueNoiseTexture::sample (bluenoise.rs:156-172) ------------------
PT_DEV void bluenoise_sample(const float* table, uint32_t size, uint32_t x, uint32_t y, uint32_t sample_index, float* sx, float* sy) {
    const uint32_t idx = (y % size) * size + (x % size);
    const float bx = table[2 * idx], by = table[2 * idx + 1];
    const float n = (float)sample_index;
    const float offset_x = fractf(n * 0.7548777f), offset_y = fractf(n * 0.5698403f);
    *sx = fractf(bx + offset_x);
    *sy = fractf(by + offset_y);
}
// Camera::generate_ray_dof (camera.rs:66-95)
PT_DEV void generate_ray_dof(const CameraDev& cam, float u, float v, float lens_u, float lens_v, V3* origin, V3* direction) {
    const float* m = cam.inv_vp;
    const float ndc_x = 2.0f * u - 1.0f, ndc_y = 1.0f - 2.0f * v;
    // Mat4 * Vec4(ndc_x, ndc_y, -1, 1) and (.., +1, 1)
    float np[4], fp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        np[r] = ((m[r] * ndc_x + m[4 + r] * ndc_y) + m[8 + r] * -1.0f) + m[12 + r] * 1.0f;
        fp[r] = ((m[r] * ndc_x + m[4 + r] * ndc_y) + m[8 + r] * 1.0f) + m[12 + r] * 1.0f;
    }
    const V3 near_point = v3(np[0], np[1], np[2]) / np[3];
    const V3 far_point = v3(fp[0], fp[1], fp[2]) / fp[3];
    const V3 dir = normalize(far_point - near_point);
    if (cam.aperture <= 0.0f) { *origin = near_point; *direction = dir; return; }
    const V3 focus_point = near_point + dir * cam.focus_distance;
    const float r = sqrtf(lens_u);
    const float theta = lens_v * PT_TAU;
    float st, ct;
    d_sincosf(theta, &st, &ct);
    const float lens_offset_x = r * ct * cam.aperture;
    const float lens_offset_y = r * st * cam.aperture;
    const V3 right = v3(cam.right[0], cam.right[1], cam.right[2]), up = v3(cam.up[0], cam.up[1], cam.up[2]);
    const V3 new_origin = near_point + right * lens_offset_x + up * lens_offset_y;
    *origin = new_origin;
    *direction = normalize(focus_point - new_origin);
}
// ray_seed (renderer/pathtracer.rs:89-94) + the sample mix (:484)
PT_DEV uint32_t path_seed(V3 o, V3 d, uint32_t sample_index) {
    const uint32_t s = __float_as_uint(o.x) + __float_as_uint(o.y) * 73856093u + __float_as_uint(o.z) * 19349663u +
                       __float_as_uint(d.x) * 83492791u;
    return s + sample_index * 2654435761u;
}

}  // namespace ptd
