// pt_scenes.cpp — seeded procedural stand-in scenes.
// The reference's assets (Bistro, Sponza, DamagedHelmet, the EXR sky) are not
// available anywhere in this pipeline, so the configs of
// BASELINE.json run on synthetic scenes of matching scale: one mesh + BLAS +
// instance per object, exactly how the reference's glTF loader lays a scene
// out (loader/gltf.rs:256-303), except "instanced_stress", which shares BLASes
// the way `Scene::new` (scene.rs:22-40) allows.
#include "pt_host_internal.h"

#include <cstdio>
#include <string>

namespace pth {
namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
    uint64_t next_u64() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float uni() { return (float)(next_u64() >> 40) * (1.0f / 16777216.0f); }
    float range(float a, float b) { return a + (b - a) * uni(); }
    uint32_t below(uint32_t n) { return (uint32_t)(next_u64() % n); }
};

// cheap lattice value noise in [0,1]
float hash3(int x, int y, int z) {
    uint32_t h = (uint32_t)x * 374761393u + (uint32_t)y * 668265263u + (uint32_t)z * 2147483647u;
    h = (h ^ (h >> 13)) * 1274126177u;
    h ^= h >> 16;
    return (float)(h & 0xFFFFFF) * (1.0f / 16777215.0f);
}
float vnoise(float x, float y, float z) {
    const float fx = std::floor(x), fy = std::floor(y), fz = std::floor(z);
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float tx = x - fx, ty = y - fy, tz = z - fz;
    tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty); tz = tz * tz * (3 - 2 * tz);
    auto L = [](float a, float b, float t) { return a + (b - a) * t; };
    const float c00 = L(hash3(ix, iy, iz), hash3(ix + 1, iy, iz), tx);
    const float c10 = L(hash3(ix, iy + 1, iz), hash3(ix + 1, iy + 1, iz), tx);
    const float c01 = L(hash3(ix, iy, iz + 1), hash3(ix + 1, iy, iz + 1), tx);
    const float c11 = L(hash3(ix, iy + 1, iz + 1), hash3(ix + 1, iy + 1, iz + 1), tx);
    return L(L(c00, c10, ty), L(c01, c11, ty), tz);
}
float fbm(V3 p, int oct) {
    float a = 0.5f, s = 0.0f;
    for (int i = 0; i < oct; ++i) { s += a * vnoise(p.x, p.y, p.z); p = p * 2.03f; a *= 0.5f; }
    return s;
}

struct MeshBuf {
    std::vector<float> pos, nrm, uv;
    std::vector<uint32_t> idx;
    bool with_normals = true, with_uvs = true;
    uint32_t vert(V3 p, V3 n, float u, float v) {
        const uint32_t i = (uint32_t)(pos.size() / 3);
        pos.push_back(p.x); pos.push_back(p.y); pos.push_back(p.z);
        if (with_normals) { nrm.push_back(n.x); nrm.push_back(n.y); nrm.push_back(n.z); }
        if (with_uvs) { uv.push_back(u); uv.push_back(v); }
        return i;
    }
    void tri(uint32_t a, uint32_t b, uint32_t c) { idx.push_back(a); idx.push_back(b); idx.push_back(c); }
    size_t ntri() const { return idx.size() / 3; }
};

// Parametric patch p(s,t), s,t in [0,1], nu x nv quads, CCW seen from +normal.
template <class F>
void add_patch(MeshBuf& m, int nu, int nv, F f, float uv_scale = 1.0f) {
    const uint32_t base = (uint32_t)(m.pos.size() / 3);
    const float e = 1e-3f;
    for (int j = 0; j <= nv; ++j)
        for (int i = 0; i <= nu; ++i) {
            const float s = (float)i / nu, t = (float)j / nv;
            const V3 p = f(s, t);
            V3 n = cross(f(s + e, t) - f(s - e, t), f(s, t + e) - f(s, t - e));
            const float l = length(n);
            n = l > 0 ? n / l : v3(0, 1, 0);
            m.vert(p, n, s * uv_scale, t * uv_scale);
        }
    for (int j = 0; j < nv; ++j)
        for (int i = 0; i < nu; ++i) {
            const uint32_t a = base + (uint32_t)(j * (nu + 1) + i), b = a + 1, c = a + (uint32_t)(nu + 1), d = c + 1;
            m.tri(a, b, d);
            m.tri(a, d, c);
        }
}

void add_quad(MeshBuf& m, V3 o, V3 ex, V3 ey, int n = 1, float uv_scale = 1.0f) {
    add_patch(m, n, n, [=](float s, float t) { return o + ex * s + ey * t; }, uv_scale);
}

// axis-aligned box [lo,hi], outward faces, each face n x n quads, optional displacement
template <class D>
void add_box(MeshBuf& m, V3 lo, V3 hi, int n, D disp, bool top = true, bool bottom = true) {
    const V3 d = hi - lo;
    auto face = [&](V3 o, V3 ex, V3 ey) {
        const V3 nn = normalize(cross(ex, ey));
        add_patch(m, n, n, [=](float s, float t) { const V3 p = o + ex * s + ey * t; return p + nn * disp(p); }, 4.0f);
    };
    face(v3(lo.x, lo.y, hi.z), v3(d.x, 0, 0), v3(0, d.y, 0));   // +z
    face(v3(hi.x, lo.y, lo.z), v3(-d.x, 0, 0), v3(0, d.y, 0));  // -z
    face(v3(hi.x, lo.y, hi.z), v3(0, 0, -d.z), v3(0, d.y, 0));  // +x
    face(v3(lo.x, lo.y, lo.z), v3(0, 0, d.z), v3(0, d.y, 0));   // -x
    if (top) face(v3(lo.x, hi.y, hi.z), v3(d.x, 0, 0), v3(0, 0, -d.z));
    if (bottom) face(v3(lo.x, lo.y, lo.z), v3(d.x, 0, 0), v3(0, 0, d.z));
}
float no_disp(V3) { return 0.0f; }

void add_cylinder(MeshBuf& m, V3 base, float r0, float r1, float h, int seg, int rings, float flute = 0.0f) {
    add_patch(m, seg, rings, [=](float s, float t) {
        const float a = -s * 2.0f * PI_F;
        const float r = (r0 + (r1 - r0) * t) * (1.0f + flute * std::cos(a * 12.0f));
        return base + v3(r * std::cos(a), h * t, r * std::sin(a));
    }, 2.0f);
}

template <class D>
void add_sphere(MeshBuf& m, V3 c, float r, int seg, int rings, D disp) {
    add_patch(m, seg, rings, [=](float s, float t) {
        const float a = -s * 2.0f * PI_F, th = (t * 0.998f + 0.001f) * PI_F;
        const V3 d = v3(std::sin(th) * std::cos(a), -std::cos(th), std::sin(th) * std::sin(a));
        return c + d * (r * (1.0f + disp(d)));
    }, 2.0f);
}

void add_arch(MeshBuf& m, V3 c, float r, float thick, float depth, int seg) {
    // half-ring in the xy plane extruded along z (front, back, inner, outer)
    auto ring = [=](float rr, float a, float z) { return c + v3(rr * std::cos(a), rr * std::sin(a), z); };
    add_patch(m, seg, 2, [=](float s, float t) { return ring(r + thick * t, PI_F * (1 - s), depth * 0.5f); });
    add_patch(m, seg, 2, [=](float s, float t) { return ring(r + thick * t, PI_F * s, -depth * 0.5f); });
    add_patch(m, seg, 2, [=](float s, float t) { return ring(r, PI_F * s, depth * (0.5f - t)); });
    add_patch(m, seg, 2, [=](float s, float t) { return ring(r + thick, PI_F * (1 - s), depth * (0.5f - t)); });
}

std::vector<float> make_texture(uint32_t w, uint32_t h, int kind, uint64_t seed) {
    // kind 0: base colour (mottled tiles), 1: metallic-roughness (G=rough, B=metal), 2: normal map, 3: emissive spots
    std::vector<float> t((size_t)w * h * 3);
    Rng r(seed);
    const V3 ca = v3(r.range(0.2f, 0.9f), r.range(0.2f, 0.9f), r.range(0.2f, 0.9f));
    const V3 cb = v3(r.range(0.1f, 0.8f), r.range(0.1f, 0.8f), r.range(0.1f, 0.8f));
    const float freq = r.range(4.0f, 16.0f);
    const float zs = (float)(seed % 1000);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const float u = (x + 0.5f) / w, v = (y + 0.5f) / h;
            const float n = fbm(v3(u * freq, v * freq, zs), 4);
            const int chk = ((int)(u * 8) + (int)(v * 8)) & 1;
            V3 c;
            if (kind == 0) c = lerp(ca, cb, n) * (chk ? 1.0f : 0.8f);
            else if (kind == 1) c = v3(1.0f, 0.15f + 0.8f * n, (n > 0.62f) ? 1.0f : 0.0f);
            else if (kind == 2) {
                const float e = 1.0f / w;
                const float dx = fbm(v3((u + e) * freq, v * freq, zs), 4) - n;
                const float dy = fbm(v3(u * freq, (v + e) * freq, zs), 4) - n;
                const V3 nn = normalize(v3(-dx * 40.0f, -dy * 40.0f, 1.0f));
                c = nn * 0.5f + splat(0.5f);
            } else c = (n > 0.7f) ? ca * 4.0f : splat(0.0f);
            float* p = &t[((size_t)y * w + x) * 3];
            p[0] = c.x; p[1] = c.y; p[2] = c.z;
        }
    return t;
}

// analytic HDR sky: horizon gradient + bright sun disc so the CDFs are non-trivial
std::vector<float> make_sky(uint32_t w, uint32_t h) {
    std::vector<float> t((size_t)w * h * 3);
    const V3 sun_dir = normalize(v3(0.5f, 1.0f, 0.3f));
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const float u = (x + 0.5f) / w, v = (y + 0.5f) / h;
            // TextureSky::uv_to_direction (sky/texture.rs:120-125)
            const float phi = (u - 0.5f) * 2.0f * PI_F, theta = (0.5f - v) * PI_F;
            const V3 d = v3(std::cos(theta) * std::sin(phi), std::sin(theta), std::cos(theta) * std::cos(phi));
            const float up = d.y;
            V3 c = up > 0 ? lerp(v3(0.9f, 0.95f, 1.0f), v3(0.25f, 0.45f, 0.9f), std::sqrt(up))
                          : lerp(v3(0.35f, 0.33f, 0.3f), v3(0.1f, 0.1f, 0.1f), std::sqrt(-up));
            const float cs = dot(d, sun_dir);
            if (cs > 0.9995f) c = c + v3(5.0e4f, 4.7e4f, 4.2e4f);
            else if (cs > 0.99f) c = c + v3(8.0f, 7.0f, 5.0f) * ((cs - 0.99f) / 0.0095f);
            float* p = &t[((size_t)y * w + x) * 3];
            p[0] = c.x; p[1] = c.y; p[2] = c.z;
        }
    return t;
}

struct SceneGen {
    std::unique_ptr<SceneData> sc = std::make_unique<SceneData>();
    Rng rng;
    float detail;
    uint32_t flags;
    explicit SceneGen(uint64_t seed, float d, uint32_t f) : rng(seed), detail(d), flags(f) {}

    int tess(int n) const { int v = (int)std::lround(n * std::sqrt(detail)); return v < 1 ? 1 : v; }

    uint32_t material(V3 base, float rough, float metal, V3 emissive = v3(0, 0, 0)) {
        PtxMaterial m;
        pth_material_default(&m);
        m.base_color_factor[0] = base.x; m.base_color_factor[1] = base.y; m.base_color_factor[2] = base.z;
        m.roughness_factor = rough; m.metallic_factor = metal;
        m.emissive_factor[0] = emissive.x; m.emissive_factor[1] = emissive.y; m.emissive_factor[2] = emissive.z;
        sc->materials.push_back(m);
        return (uint32_t)sc->materials.size() - 1;
    }
    uint32_t texture(uint32_t w, uint32_t h, int kind) {
        auto t = std::make_unique<TextureData>();
        t->width = w; t->height = h;
        t->texels = make_texture(w, h, kind, rng.next_u64());
        sc->textures.push_back(std::move(t));
        return (uint32_t)sc->textures.size() - 1;
    }
    uint32_t add(MeshBuf& m, uint32_t mat, const M4& xf = M4::identity()) {
        const int32_t b = sc->add_mesh(std::move(m.pos), std::move(m.nrm), std::move(m.uv), std::move(m.idx), mat);
        sc->instance_data.emplace_back((uint32_t)b, xf);
        return (uint32_t)b;
    }
    void lighting() {
        if (flags & 1u) {  // app lighting (app.rs:121-132) with a procedural stand-in for default.exr
            uint32_t w = 512, h = 256;
            if (detail >= 0.5f) { w = 2048; h = 1024; }
            sc->sky = SkyData();
            sc->sky.texture.width = w; sc->sky.texture.height = h;
            sc->sky.texture.texels = make_sky(w, h);
            sc->set_sun(normalize(v3(-0.5f, -1.0f, -0.3f)), v3(1.0f, 0.95f, 0.85f) * 8.0f);
        } else {  // --benchmark (main.rs:109-112)
            sc->sky = SkyData();
            sc->sky.texture.solid = v3(0.5f, 0.6f, 0.8f);
        }
    }
    void apply_sss_fraction() {
        if (!(flags & 2u)) return;
        for (size_t i = 0; i < sc->materials.size(); ++i)
            if (i % 10 == 3 && sc->materials[i].transmission == 0.0f) sc->materials[i].subsurface = 1.0f;
    }
};

// ---- C1: Cornell box, 36 triangles, 8 meshes -------------------------------
std::unique_ptr<SceneData> gen_cornell(uint64_t seed, float detail, uint32_t flags) {
    SceneGen g(seed, detail, flags);
    const uint32_t white = g.material(v3(0.73f, 0.73f, 0.73f), 1.0f, 0.0f);
    const uint32_t red = g.material(v3(0.65f, 0.05f, 0.05f), 1.0f, 0.0f);
    const uint32_t green = g.material(v3(0.12f, 0.45f, 0.15f), 1.0f, 0.0f);
    const uint32_t light = g.material(v3(0.0f, 0.0f, 0.0f), 1.0f, 0.0f, v3(15, 15, 15));
    const uint32_t metal = g.material(v3(0.9f, 0.85f, 0.7f), 0.25f, 1.0f);
    auto wall = [&](V3 o, V3 ex, V3 ey, uint32_t mat, bool normals) {
        MeshBuf m; m.with_normals = normals; m.with_uvs = false;
        add_quad(m, o, ex, ey);
        g.add(m, mat);
    };
    wall(v3(-1, 0, 1), v3(2, 0, 0), v3(0, 0, -2), white, true);     // floor (normal +y)
    wall(v3(-1, 2, -1), v3(2, 0, 0), v3(0, 0, 2), white, false);    // ceiling (normal -y)
    wall(v3(-1, 0, -1), v3(2, 0, 0), v3(0, 2, 0), white, true);     // back (+z)
    wall(v3(-1, 0, 1), v3(0, 0, -2), v3(0, 2, 0), red, false);      // left (+x)
    wall(v3(1, 0, -1), v3(0, 0, 2), v3(0, 2, 0), green, true);      // right (-x)
    wall(v3(-0.3f, 1.995f, -0.3f), v3(0.6f, 0, 0), v3(0, 0, 0.6f), light, false);
    {
        MeshBuf m; m.with_uvs = false;
        add_box(m, v3(-0.3f, 0, -0.3f), v3(0.3f, 1.2f, 0.3f), 1, no_disp);
        g.add(m, white, mul(translation(v3(-0.35f, 0, -0.3f)), rotation_axis(v3(0, 1, 0), 0.3f)));
    }
    {
        MeshBuf m; m.with_normals = false; m.with_uvs = false;
        add_box(m, v3(-0.3f, 0, -0.3f), v3(0.3f, 0.6f, 0.3f), 1, no_disp);
        g.add(m, metal, mul(translation(v3(0.4f, 0, 0.35f)), rotation_axis(v3(0, 1, 0), -0.4f)));
    }
    g.lighting();
    return std::move(g.sc);
}

// ---- C1b: helmet-like textured object --------------------------------------
std::unique_ptr<SceneData> gen_helmet(uint64_t seed, float detail, uint32_t flags) {
    SceneGen g(seed, detail, flags);
    const uint32_t ts = detail >= 0.5f ? 2048 : (detail >= 0.1f ? 512 : 128);
    PtxMaterial pm;
    pth_material_default(&pm);
    pm.base_color_texture = (int32_t)g.texture(ts, ts, 0);
    pm.metallic_roughness_texture = (int32_t)g.texture(ts, ts, 1);
    pm.normal_texture = (int32_t)g.texture(ts, ts, 2);
    pm.emissive_texture = (int32_t)g.texture(ts, ts, 3);
    pm.emissive_factor[0] = pm.emissive_factor[1] = pm.emissive_factor[2] = 1.0f;
    pm.normal_scale = 0.8f;
    g.sc->materials.push_back(pm);
    const uint32_t ground = g.material(v3(0.5f, 0.5f, 0.5f), 0.9f, 0.0f);
    {
        MeshBuf m;
        const int n = g.tess(186);  // ~70k triangles at detail 1
        const float zs = (float)(seed % 97);
        add_sphere(m, v3(0, 0.8f, 0), 0.7f, n, n, [=](V3 d) { return 0.12f * fbm(d * 3.0f + splat(zs), 4); });
        g.add(m, 0, rotation_axis(v3(0, 1, 0), 0.5f));
    }
    { MeshBuf m; add_quad(m, v3(-6, 0, 6), v3(12, 0, 0), v3(0, 0, -12), 8, 8.0f); g.add(m, ground); }
    g.lighting();
    g.apply_sss_fraction();
    return std::move(g.sc);
}

// ---- C2: Sponza-like atrium, ~262k triangles, ~100 meshes ------------------
std::unique_ptr<SceneData> gen_sponza(uint64_t seed, float detail, uint32_t flags) {
    SceneGen g(seed, detail, flags);
    const uint32_t ts = detail >= 0.5f ? 1024 : (detail >= 0.1f ? 256 : 64);
    std::vector<uint32_t> tex_base, tex_mr, tex_n;
    for (int i = 0; i < 6; ++i) { tex_base.push_back(g.texture(ts, ts, 0)); tex_mr.push_back(g.texture(ts, ts, 1)); tex_n.push_back(g.texture(ts, ts, 2)); }
    std::vector<uint32_t> mats;
    for (int i = 0; i < 25; ++i) {
        PtxMaterial m;
        pth_material_default(&m);
        const int k = i % 6;
        if (i % 3 != 2) m.base_color_texture = (int32_t)tex_base[k];
        if (i % 4 == 0) m.metallic_roughness_texture = (int32_t)tex_mr[k];
        if (i % 2 == 0) m.normal_texture = (int32_t)tex_n[k];
        m.base_color_factor[0] = g.rng.range(0.5f, 1.0f); m.base_color_factor[1] = g.rng.range(0.5f, 1.0f); m.base_color_factor[2] = g.rng.range(0.5f, 1.0f);
        m.roughness_factor = g.rng.range(0.3f, 1.0f);
        m.metallic_factor = (i % 4 == 0) ? 1.0f : 0.0f;
        g.sc->materials.push_back(m);
        mats.push_back((uint32_t)g.sc->materials.size() - 1);
    }
    auto pick = [&]() { return mats[g.rng.below((uint32_t)mats.size())]; };
    // floor, ceiling beams, long walls
    { MeshBuf m; const int n = g.tess(96); add_patch(m, n, n / 2, [](float s, float t) { return v3(-20 + 40 * s, 0.02f * std::sin(s * 90) * std::sin(t * 40), 8 - 16 * t); }, 16.0f); g.add(m, mats[0]); }
    for (int side = 0; side < 2; ++side) {
        MeshBuf m; const int n = g.tess(64); const float z = side ? 8.0f : -8.0f; const float sg = side ? -1.0f : 1.0f;
        add_patch(m, n, n / 2, [=](float s, float t) { const V3 p = v3(side ? 20 - 40 * s : -20 + 40 * s, 12 * t, z); return p + v3(0, 0, sg * 0.15f * fbm(v3(p.x, p.y, 0.0f), 3)); }, 12.0f);
        g.add(m, mats[1 + side]);
    }
    for (int e = 0; e < 2; ++e) {
        MeshBuf m; const float x = e ? 20.0f : -20.0f;
        add_quad(m, v3(x, 0, e ? 8.0f : -8.0f), v3(0, 0, e ? -16.0f : 16.0f), v3(0, 12, 0), g.tess(8), 6.0f);
        g.add(m, mats[3]);
    }
    // two storeys of columns + arches along both sides
    const int ncol = 11;
    for (int level = 0; level < 2; ++level)
        for (int side = 0; side < 2; ++side)
            for (int i = 0; i < ncol; ++i) {
                const float x = -17.5f + 3.5f * i, z = side ? 4.5f : -4.5f, y0 = level * 5.5f;
                { MeshBuf m; add_cylinder(m, v3(0, 0, 0), 0.32f, 0.26f, 4.0f, g.tess(48), g.tess(40), 0.04f);
                  g.add(m, pick(), translation(v3(x, y0, z))); }
                if (i + 1 < ncol) { MeshBuf m; add_arch(m, v3(0, 0, 0), 1.45f, 0.35f, 0.6f, g.tess(40));
                  g.add(m, pick(), translation(v3(x + 1.75f, y0 + 4.0f, z))); }
            }
    // hanging cloth banners
    for (int i = 0; i < 8; ++i) {
        MeshBuf m; const int n = g.tess(56); const float ph = g.rng.range(0, 6.28f);
        add_patch(m, n, n, [=](float s, float t) { return v3(-1.2f + 2.4f * s, -3.5f * t, 0.25f * std::sin(s * 9 + ph) * t + 0.1f * std::sin(t * 14 + ph)); }, 2.0f);
        g.add(m, pick(), mul(translation(v3(-14.0f + 4.0f * i, 10.0f, (i & 1) ? 1.5f : -1.5f)), rotation_axis(v3(0, 1, 0), g.rng.range(-0.3f, 0.3f))));
    }
    // a few props (vases = displaced spheres, plinths)
    for (int i = 0; i < 10; ++i) {
        const float x = g.rng.range(-15, 15), z = g.rng.range(-2.5f, 2.5f);
        { MeshBuf m; add_box(m, v3(-0.4f, 0, -0.4f), v3(0.4f, 0.8f, 0.4f), g.tess(6), no_disp); g.add(m, pick(), mul(translation(v3(x, 0, z)), rotation_axis(v3(0, 1, 0), g.rng.range(0, 3.0f)))); }
        { MeshBuf m; const int n = g.tess(40); add_sphere(m, v3(0, 0, 0), 0.45f, n, n, [](V3 d) { return 0.25f * d.y * d.y; }); g.add(m, pick(), translation(v3(x, 1.25f, z))); }
    }
    g.lighting();
    g.apply_sss_fraction();
    return std::move(g.sc);
}

// ---- C3/C5: Bistro-like street, ~2.8M triangles, ~1000 meshes --------------
std::unique_ptr<SceneData> gen_bistro(uint64_t seed, float detail, uint32_t flags) {
    SceneGen g(seed, detail, flags);
    const uint32_t ts = detail >= 0.5f ? 1024 : (detail >= 0.1f ? 256 : 64);
    std::vector<uint32_t> tex_base, tex_mr, tex_n, tex_e;
    for (int i = 0; i < 12; ++i) { tex_base.push_back(g.texture(ts, ts, 0)); tex_mr.push_back(g.texture(ts, ts, 1)); tex_n.push_back(g.texture(ts, ts, 2)); }
    for (int i = 0; i < 2; ++i) tex_e.push_back(g.texture(ts / 2, ts / 2, 3));
    std::vector<uint32_t> mats;
    for (int i = 0; i < 130; ++i) {
        PtxMaterial m;
        pth_material_default(&m);
        const int k = i % 12;
        if (i % 5 != 4) m.base_color_texture = (int32_t)tex_base[k];
        if (i % 3 == 0) m.metallic_roughness_texture = (int32_t)tex_mr[k];
        if (i % 2 == 1) m.normal_texture = (int32_t)tex_n[k];
        if (i % 26 == 7) { m.emissive_texture = (int32_t)tex_e[i & 1]; m.emissive_factor[0] = m.emissive_factor[1] = m.emissive_factor[2] = 1.0f; }
        m.base_color_factor[0] = g.rng.range(0.4f, 1.0f); m.base_color_factor[1] = g.rng.range(0.4f, 1.0f); m.base_color_factor[2] = g.rng.range(0.4f, 1.0f);
        m.roughness_factor = g.rng.range(0.2f, 1.0f);
        m.metallic_factor = (i % 3 == 0) ? 1.0f : 0.0f;
        if (i % 13 == 5) { m.transmission = 1.0f; m.roughness_factor = 0.05f; }   // glass (Material::glass)
        g.sc->materials.push_back(m);
        mats.push_back((uint32_t)g.sc->materials.size() - 1);
    }
    const uint32_t lamp = g.material(v3(0, 0, 0), 1.0f, 0.0f, v3(12, 10, 7));
    auto pick = [&]() { return mats[g.rng.below((uint32_t)mats.size())]; };
    auto opaque = [&]() { uint32_t m; do { m = pick(); } while (g.sc->materials[m].transmission > 0); return m; };

    // cobbled street
    { MeshBuf m; const int n = g.tess(300);
      add_patch(m, n, n, [](float s, float t) { const float x = -40 + 80 * s, z = 40 - 80 * t; return v3(x, 0.03f * vnoise(x * 6, 0.0f, z * 6), z); }, 40.0f);
      g.add(m, mats[0]); }
    // two rows of building facades along a Y-shaped street
    const int nb = 30;
    for (int i = 0; i < 2 * nb; ++i) {
        const int side = i & 1, k = i >> 1;
        const float w = g.rng.range(5.0f, 7.5f), h = g.rng.range(7.0f, 16.0f), dpt = g.rng.range(5.0f, 8.0f);
        const float x = -38.0f + (76.0f / nb) * k + g.rng.range(-0.3f, 0.3f);
        const float z = side ? (7.0f + 0.08f * x) : (-7.0f - 0.08f * x) - dpt;
        const float zs = g.rng.range(0, 100);
        MeshBuf m; const int n = g.tess(50);
        add_box(m, v3(0, 0, 0), v3(w, h, dpt), n, [=](V3 p) {
            // window reliefs + plaster noise
            const float wx = std::fabs(std::sin(p.x * 2.6f)) , wy = std::fabs(std::sin(p.y * 2.1f));
            const float relief = (wx > 0.75f && wy > 0.6f) ? -0.18f : 0.0f;
            return relief + 0.05f * fbm(v3(p.x + zs, p.y, p.z) * 2.0f, 3);
        }, true, false);
        g.add(m, opaque(), mul(translation(v3(x, 0, z)), rotation_axis(v3(0, 1, 0), g.rng.range(-0.04f, 0.04f))));
    }
    // trees
    for (int i = 0; i < 90; ++i) {
        const float x = g.rng.range(-36, 36), z = g.rng.range(-5.0f, 5.0f) * (g.rng.uni() < 0.5f ? 1.0f : 0.6f);
        const float s = g.rng.range(0.8f, 1.5f), zs = g.rng.range(0, 50);
        { MeshBuf m; add_cylinder(m, v3(0, 0, 0), 0.18f, 0.1f, 2.6f, g.tess(16), g.tess(8)); g.add(m, opaque(), mul(translation(v3(x, 0, z)), scaling(v3(s, s, s)))); }
        { MeshBuf m; const int n = g.tess(52); add_sphere(m, v3(0, 0, 0), 1.3f, n, n, [=](V3 d) { return 0.45f * fbm(d * 2.5f + splat(zs), 4); });
          g.add(m, opaque(), mul(translation(v3(x, 3.4f * s, z)), mul(rotation_axis(v3(0.2f, 1, 0.1f), g.rng.range(0, 6.0f)), scaling(v3(s, s * 1.1f, s))))); }
    }
    // street lamps
    for (int i = 0; i < 60; ++i) {
        const float x = -37.0f + 1.25f * i, z = (i & 1) ? 5.6f : -5.6f;
        { MeshBuf m; add_cylinder(m, v3(0, 0, 0), 0.07f, 0.05f, 4.2f, g.tess(12), g.tess(10)); g.add(m, mats[3], translation(v3(x, 0, z))); }
        { MeshBuf m; const int n = g.tess(14); add_sphere(m, v3(0, 0, 0), 0.22f, n, n, [](V3) { return 0.0f; }); g.add(m, (i % 3 == 0) ? lamp : mats[5], translation(v3(x, 4.35f, z))); }
    }
    // bistro furniture: tables (top + 4 legs as one mesh) and chairs
    for (int i = 0; i < 160; ++i) {
        const float x = g.rng.range(-34, 34), z = g.rng.range(2.0f, 5.0f) * (g.rng.uni() < 0.5f ? 1.0f : -1.0f);
        const float rot = g.rng.range(0, 6.28f);
        { MeshBuf m; const int n = g.tess(4);
          add_box(m, v3(-0.45f, 0.72f, -0.45f), v3(0.45f, 0.76f, 0.45f), n, no_disp);
          for (int l = 0; l < 4; ++l) { const float lx = (l & 1) ? 0.38f : -0.38f, lz = (l & 2) ? 0.38f : -0.38f; add_box(m, v3(lx - 0.03f, 0, lz - 0.03f), v3(lx + 0.03f, 0.72f, lz + 0.03f), 1, no_disp); }
          g.add(m, pick(), mul(translation(v3(x, 0, z)), rotation_axis(v3(0, 1, 0), rot))); }
        for (int c = 0; c < 3; ++c) {
            MeshBuf m; const float a = rot + c * 2.1f;
            add_box(m, v3(-0.2f, 0.42f, -0.2f), v3(0.2f, 0.46f, 0.2f), g.tess(2), no_disp);
            add_box(m, v3(-0.2f, 0.46f, 0.17f), v3(0.2f, 0.9f, 0.2f), g.tess(2), no_disp);
            for (int l = 0; l < 4; ++l) { const float lx = (l & 1) ? 0.17f : -0.17f, lz = (l & 2) ? 0.17f : -0.17f; add_box(m, v3(lx - 0.02f, 0, lz - 0.02f), v3(lx + 0.02f, 0.42f, lz + 0.02f), 1, no_disp); }
            g.add(m, opaque(), mul(translation(v3(x + 0.75f * std::cos(a), 0, z + 0.75f * std::sin(a))), rotation_axis(v3(0, 1, 0), -a)));
        }
    }
    // awnings (cloth)
    for (int i = 0; i < 40; ++i) {
        MeshBuf m; const int n = g.tess(36); const float ph = g.rng.range(0, 6.28f);
        add_patch(m, n, n, [=](float s, float t) { return v3(-1.5f + 3 * s, -0.8f * t + 0.05f * std::sin(s * 20 + ph), 1.6f * t + 0.04f * std::sin(t * 9 + ph)); }, 3.0f);
        const int side = i & 1; const float x = -36.0f + 1.8f * i;
        g.add(m, opaque(), mul(translation(v3(x, 3.4f, side ? 6.9f + 0.08f * x - 1.6f : -6.9f - 0.08f * x)), rotation_axis(v3(0, 1, 0), side ? PI_F : 0.0f)));
    }
    // bottles / glasses / ornaments (glass, metal, candidates for SSS)
    for (int i = 0; i < 140; ++i) {
        MeshBuf m; const int n = g.tess(24);
        add_sphere(m, v3(0, 0, 0), 0.12f, n, n, [](V3 d) { return 0.35f * d.y; });
        g.add(m, pick(), translation(v3(g.rng.range(-34, 34), 0.9f, g.rng.range(-5, 5))));
    }
    g.lighting();
    g.apply_sss_fraction();
    return std::move(g.sc);
}

// ---- C4: 10k instances of 4 BLASes -----------------------------------------
std::unique_ptr<SceneData> gen_instanced(uint64_t seed, float detail, uint32_t flags) {
    SceneGen g(seed, detail, flags);
    std::vector<uint32_t> mats;
    for (int i = 0; i < 8; ++i) mats.push_back(g.material(v3(g.rng.range(0.3f, 0.95f), g.rng.range(0.3f, 0.95f), g.rng.range(0.3f, 0.95f)), g.rng.range(0.2f, 1.0f), (i % 3 == 0) ? 1.0f : 0.0f));
    const int res[4] = {22, 50, 100, 158};  // ~1k, 5k, 20k, 50k triangles
    std::vector<uint32_t> blas;
    for (int b = 0; b < 4; ++b) {
        MeshBuf m; const int n = g.tess(res[b]); const float zs = (float)b * 7.0f;
        if (b == 3) add_patch(m, n, n, [](float s, float t) { const float a = s * 2 * PI_F, c = t * 2 * PI_F; const float r = 1.0f + 0.4f * std::cos(c); return v3(r * std::cos(a), 0.4f * std::sin(c), -r * std::sin(a)); });
        else add_sphere(m, v3(0, 0, 0), 1.0f, n, n, [=](V3 d) { return 0.3f * fbm(d * 2.0f + splat(zs), 3); });
        const int32_t id = g.sc->add_mesh(std::move(m.pos), std::move(m.nrm), std::move(m.uv), std::move(m.idx), mats[(size_t)b * 2]);
        blas.push_back((uint32_t)id);
    }
    const int n_inst = std::max(16, (int)std::lround(10000 * detail));
    for (int i = 0; i < n_inst; ++i) {
        const V3 axis = normalize(v3(g.rng.range(-1, 1), g.rng.range(-1, 1), g.rng.range(-1, 1) + 1e-3f));
        const float s = g.rng.range(0.5f, 2.0f);
        const V3 t = v3(g.rng.range(-100, 100), g.rng.range(-100, 100), g.rng.range(-100, 100));
        g.sc->instance_data.emplace_back(blas[g.rng.below(4)], mul(translation(t), mul(rotation_axis(axis, g.rng.range(0, 6.28f)), scaling(v3(s, s, s)))));
    }
    g.lighting();
    return std::move(g.sc);
}

// ---- small material tests ---------------------------------------------------
std::unique_ptr<SceneData> gen_material_test(uint64_t seed, float detail, uint32_t flags, int which) {
    SceneGen g(seed, detail, flags);
    const uint32_t ground = g.material(v3(0.6f, 0.6f, 0.6f), 0.8f, 0.0f);
    { MeshBuf m; add_quad(m, v3(-5, 0, 5), v3(10, 0, 0), v3(0, 0, -10), 4, 4.0f); g.add(m, ground); }
    PtxMaterial a, b;
    pth_material_default(&a); pth_material_default(&b);
    if (which == 0) {  // Material::subsurface + a second, anisotropic one
        a.subsurface = 1.0f; a.base_color_factor[0] = 0.9f; a.base_color_factor[1] = 0.7f; a.base_color_factor[2] = 0.6f;
        b.subsurface = 0.6f; b.subsurface_anisotropy = 0.4f; b.subsurface_radius[0] = 0.3f; b.subsurface_radius[1] = 0.3f; b.subsurface_radius[2] = 0.3f;
    } else {           // Material::glass + Material::metal
        a.transmission = 1.0f; a.roughness_factor = 0.0f; a.ior = 1.5f;
        b.metallic_factor = 1.0f; b.roughness_factor = 0.2f; b.base_color_factor[0] = 0.95f; b.base_color_factor[1] = 0.8f; b.base_color_factor[2] = 0.4f;
    }
    g.sc->materials.push_back(a); const uint32_t ma = (uint32_t)g.sc->materials.size() - 1;
    g.sc->materials.push_back(b); const uint32_t mb = (uint32_t)g.sc->materials.size() - 1;
    { MeshBuf m; const int n = g.tess(48); add_sphere(m, v3(0, 0, 0), 0.6f, n, n, [](V3) { return 0.0f; }); g.add(m, ma, translation(v3(-0.7f, 0.6f, 0))); }
    { MeshBuf m; const int n = g.tess(48); add_sphere(m, v3(0, 0, 0), 0.5f, n, n, [](V3 d) { return 0.1f * d.x * d.y; }); g.add(m, mb, mul(translation(v3(0.7f, 0.5f, 0.2f)), rotation_axis(v3(1, 0, 0), 0.4f))); }
    { MeshBuf m; add_box(m, v3(-0.25f, 0, -0.25f), v3(0.25f, 0.5f, 0.25f), 2, no_disp); g.add(m, ma, mul(translation(v3(0.0f, 0, 1.2f)), rotation_axis(v3(0, 1, 0), 0.6f))); }
    g.lighting();
    return std::move(g.sc);
}

}  // namespace
}  // namespace pth

using namespace pth;

extern "C" {

PthScene* pth_scene_procedural(const char* name, uint64_t seed, float detail, uint32_t flags) {
    if (!name || !(detail > 0.0f)) return nullptr;
    const std::string n(name);
    std::unique_ptr<SceneData> sc;
    if (n == "cornell") sc = gen_cornell(seed, detail, flags);
    else if (n == "helmet_like") sc = gen_helmet(seed, detail, flags);
    else if (n == "sponza_like") sc = gen_sponza(seed, detail, flags);
    else if (n == "bistro_like") sc = gen_bistro(seed, detail, flags);
    else if (n == "instanced_stress") sc = gen_instanced(seed, detail, flags);
    else if (n == "sss_test") sc = gen_material_test(seed, detail, flags, 0);
    else if (n == "glass_test") sc = gen_material_test(seed, detail, flags, 1);
    else return nullptr;
    auto* s = new PthScene();
    s->scene = std::move(sc);
    s->scene->finish();
    return s;
}

int32_t pth_scene_default_camera(const char* name, float aspect_ratio, PtxCamera* out) {
    if (!name || !out) return -1;
    const std::string n(name);
    V3 eye = v3(0.0f, 1.0f, 3.0f), at = v3(0.0f, 0.5f, 0.0f);  // main.rs:98-100
    float fov = 60.0f;
    if (n == "sponza_like") { eye = v3(-16.0f, 2.2f, 0.6f); at = v3(6.0f, 4.0f, -0.8f); }
    else if (n == "bistro_like") { eye = v3(-30.0f, 2.0f, 1.0f); at = v3(10.0f, 3.5f, -0.5f); }
    else if (n == "instanced_stress") { eye = v3(0.0f, 0.0f, 230.0f); at = v3(0.0f, 0.0f, 0.0f); }
    else if (n == "cornell") { eye = v3(0.0f, 1.0f, 3.0f); at = v3(0.0f, 1.0f, 0.0f); fov = 50.0f; }
    else if (n == "helmet_like" || n == "sss_test" || n == "glass_test") { eye = v3(0.0f, 1.0f, 3.0f); at = v3(0.0f, 0.5f, 0.0f); }
    else return -1;
    camera_new(eye, at, fov, aspect_ratio, out);
    return 0;
}

}  // extern "C"
