"""Known-answer tests that pin pieces of the CPU oracle.

The reference ships no tests or fixtures for this path (parity unpinned by the reference itself), so each
answer here is derived by hand / with Python big-ints / in float64 from the cited reference lines.
"""
import math

import numpy as np
import pytest

M64 = (1 << 64) - 1


def pcg_reference(seed, n):
    """renderer/subsurface.rs:18-34 with Python integers."""
    state = seed & 0xFFFFFFFF
    outs = []

    def step():
        nonlocal state
        old = state
        state = (old * 6364136223846793005 + 1442695040888963407) & M64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xFFFFFFFF
    step()   # Rng::new warms up once
    for _ in range(n):
        outs.append(step())
    return outs


@pytest.mark.parametrize("seed", [0, 1, 0xFFFFFFFF, 12345, 2654435761])
def test_pcg_stream(oracle_det, seed):
    u, f = oracle_det.rng_stream(seed, 64)
    ref = pcg_reference(seed, 64)
    assert u.tolist() == ref
    # next() = u32 as f32 / (u32::MAX as f32): both round to nearest, so 1.0 is reachable
    expect = (np.array(ref, dtype=np.uint64).astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
    assert np.array_equal(f, expect) and f.min() >= 0.0 and f.max() <= 1.0


def test_scalar_helpers(oracle_det):
    L = oracle_det.L
    assert abs(L.orc_mis_weight(0.37, 0.37) - 0.5) < 1e-6          # sampling.rs:34-38
    assert L.orc_mis_weight(1.0, 0.0) == pytest.approx(1.0, abs=1e-6)
    assert L.orc_mis_weight(0.0, 0.0) == 0.0                         # the +1e-10 guard
    assert L.orc_safe_inv_pdf(1e-3) == 100.0                         # sampling.rs:40-46
    assert L.orc_safe_inv_pdf(0.5) == 2.0
    assert np.array_equal(oracle_det.clamp_radiance((20, 10, 0)), np.array([10, 5, 0], dtype=np.float32))   # :48-59
    assert np.array_equal(oracle_det.clamp_radiance((1, 2, 3)), np.array([1, 2, 3], dtype=np.float32))
    assert np.array_equal(oracle_det.clamp_radiance((np.nan, 1, 1)), np.zeros(3, dtype=np.float32))
    assert np.array_equal(oracle_det.clamp_radiance((np.inf, 1, 1)), np.zeros(3, dtype=np.float32))
    # fresnel_dielectric (brdf.rs:43-46): r0 = ((1-1.5)/(1+1.5))^2 = 0.04 at normal incidence, 1 at grazing
    assert L.orc_fresnel_dielectric(1.0, 1.5) == pytest.approx(0.04, rel=1e-6)
    assert L.orc_fresnel_dielectric(0.0, 1.5) == pytest.approx(1.0, rel=1e-6)


def test_evaluate_brdf_lambert_limit(oracle_det):
    """brdf.rs:81-102 with f0 = 0: only the diffuse lobe survives: diffuse_color * (1 - F) / pi * n.l, F = (1-v.h)^5."""
    n = np.array([0, 0, 1.0])
    v = np.array([0.3, 0.1, 0.9]); v /= np.linalg.norm(v)
    l = np.array([-0.2, 0.4, 0.7]); l /= np.linalg.norm(l)
    hvec = (v + l) / np.linalg.norm(v + l)
    F = (1 - v @ hvec) ** 5
    expect = np.array([0.8, 0.5, 0.2]) * (1 - F) / math.pi * (n @ l)
    got = oracle_det.evaluate_brdf(v, l, n, 0.25, (0, 0, 0), (0.8, 0.5, 0.2))
    assert np.allclose(got, expect, rtol=2e-5)
    assert np.array_equal(oracle_det.evaluate_brdf(v, -l, n, 0.25, (0, 0, 0), (1, 1, 1)), np.zeros(3, dtype=np.float32))


def test_solid_sky_cdfs_and_uniform_pdf(pkg, oracle_det):
    """--benchmark's sky (main.rs:109-110): 1x1 CDFs [0,1]; total_weight = lum(0.5,0.6,0.8) * sin(pi/2)."""
    b = pkg.SceneBuilder()
    b.set_sky_solid(0.5, 0.6, 0.8)
    sc = b.build()
    sky = sc.desc.contents.sky
    assert (sky.width, sky.height) == (1, 1)
    assert [sky.marginal_cdf[i] for i in range(2)] == [0.0, 1.0]
    assert [sky.conditional_cdfs[i] for i in range(2)] == [0.0, 1.0]
    assert sky.total_weight == pytest.approx(0.2126 * 0.5 + 0.7152 * 0.6 + 0.0722 * 0.8, rel=1e-6)   # 0.59318
    osc = oracle_det.scene(sc)
    d, rad, pdf = osc.sky_sample_direction(0.25, 0.5)
    assert np.allclose(rad, (0.5, 0.6, 0.8)) and abs(np.linalg.norm(d) - 1) < 1e-5
    # pdf (sky/texture.rs:190-210) for a 1x1 map: 1*1*1 / (2 pi^2 sin(pi v)); sampled v = 0.25 -> sin(pi/4)
    assert pdf == pytest.approx(1.0 / (2 * math.pi ** 2 * math.sin(math.pi * 0.25)), rel=1e-4)
    assert np.allclose(osc.sky_sample((0.0, 1.0, 0.0)), (0.5, 0.6, 0.8))


def test_sky_image_importance_sampling(pkg, oracle_libm):
    """A 4x2 map with one bright texel: the CDF search must land in it and pdf * solid angle must integrate to 1."""
    tex = np.full((2, 4, 3), 0.01, dtype=np.float32)
    tex[0, 2] = 100.0
    b = pkg.SceneBuilder()
    b.set_sky_image(tex)
    sc = b.build()
    sky = sc.desc.contents.sky
    marg = np.array([sky.marginal_cdf[i] for i in range(3)])
    assert marg[0] == 0 and marg[2] == pytest.approx(1.0, rel=1e-6) and np.all(np.diff(marg) > 0)
    cond0 = np.array([sky.conditional_cdfs[i] for i in range(5)])
    assert cond0[-1] == pytest.approx(1.0, rel=1e-6) and np.all(np.diff(cond0) >= 0)
    osc = oracle_libm.scene(sc)
    hits = 0
    rng = np.random.default_rng(0)
    for _ in range(200):
        d, rad, pdf = osc.sky_sample_direction(float(rng.random()), float(rng.random()))
        assert pdf > 0 and abs(np.linalg.norm(d) - 1) < 1e-4
        hits += rad.max() > 1.0
    assert hits > 150   # nearly all samples go to (or bilinearly next to) the bright texel
    # integrate pdf over the sphere through the same (u,v)->direction map: sum pdf * sin(theta) dtheta dphi = 1
    n = 64
    tot = 0.0
    for j in range(n):
        for i in range(2 * n):
            u, v = (i + 0.5) / (2 * n), (j + 0.5) / n
            phi, theta = (u - 0.5) * 2 * math.pi, (0.5 - v) * math.pi
            dvec = (math.cos(theta) * math.cos(phi), math.sin(theta), math.cos(theta) * math.sin(phi))   # inverse of direction_to_uv
            tot += osc.sky_pdf(dvec) * math.cos(theta) * (math.pi / n) * (2 * math.pi / (2 * n))
    assert tot == pytest.approx(1.0, rel=2e-2)


def test_directional_light_constants(pkg):
    """light.rs:13-26 with the app's sun (app.rs:129-132)."""
    b = pkg.SceneBuilder()
    b.set_sky_solid(1, 1, 1)
    b.set_sun((-0.5, -1.0, -0.3), (8.0, 7.6, 6.8))
    sun = b.build().desc.contents.sun
    ang = math.radians(0.25)
    assert sun.enabled == 1
    assert sun.cos_angle == pytest.approx(math.cos(ang), rel=1e-7)
    assert sun.pdf == pytest.approx(1.0 / (2 * math.pi * (1 - math.cos(ang))), rel=2e-2)   # ~1.67e4; f32 cancellation in 1-cos
    d = np.array(list(sun.direction))
    assert np.allclose(d, np.array([-0.5, -1.0, -0.3]) / np.linalg.norm([-0.5, -1.0, -0.3]), atol=1e-6)


def test_r2_blue_noise_offsets(pkg, oracle_det):
    """bluenoise.rs:156-172: fract(base + fract(n * a)) with a = (0.7548777, 0.5698403)."""
    sc = pkg.Scene.procedural("cornell")
    osc = oracle_det.scene(sc)
    table = pkg.host.bluenoise_generate(64, 12345)
    for (x, y) in ((0, 0), (5, 9), (63, 63), (64, 70), (1919, 1079)):
        base = table[(y % 64) * 64 + (x % 64)]
        for n in range(4):
            off = np.array([np.float32(n) * np.float32(0.7548777), np.float32(n) * np.float32(0.5698403)], dtype=np.float32)
            off = off - np.trunc(off)
            e = (base + off).astype(np.float32)
            e = e - np.trunc(e)
            assert np.array_equal(osc.bluenoise_sample(0, x, y, n), e.astype(np.float32))
    assert np.array_equal(osc.bluenoise_sample(0, 3, 4, 0), table[4 * 64 + 3])   # n = 0: the table itself


def single_triangle_scene(pkg, transform=None):
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material())
    m = b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]])
    b.add_instance(m, transform)
    b.set_sky_solid(0.5, 0.6, 0.8)
    return b.build()


def test_single_triangle_analytic(pkg, oracle_det):
    """intersect_triangle (shape/triangle_mesh.rs:63-100): ray straight down -z onto the z=0 triangle."""
    abi = pkg._abi
    osc = oracle_det.scene(single_triangle_scene(pkg))
    rays = np.zeros(4, dtype=abi.ray_dtype())
    rays["origin"] = [(0.25, 0.5, 2.0), (0.9, 0.9, 2.0), (0.25, 0.5, -3.0), (0.25, 0.5, 0.0005)]
    rays["direction"] = [(0, 0, -1), (0, 0, -1), (0, 0, 1), (0, 0, -1)]
    rays["t_max"] = np.inf
    h = osc.trace_closest(rays)
    assert (h["inst_idx"][0], h["tri_idx"][0]) == (0, 0)
    assert h["t"][0] == np.float32(2.0) and h["u"][0] == np.float32(0.25) and h["v"][0] == np.float32(0.5)
    assert h["inst_idx"][1] == abi.PTX_MISS                      # u + v > 1
    assert h["inst_idx"][2] == 0 and h["t"][2] == np.float32(3.0)   # two-sided
    assert h["inst_idx"][3] == abi.PTX_MISS                      # t = 0.0005 < t_min = RAY_EPSILON
    assert osc.trace_any(rays).tolist() == [1, 0, 1, 0]
    # hit data: no normals/uvs in the mesh -> face normal and barycentric uv (triangle_mesh.rs:113-135)
    hd = osc.trace_hitdata(rays[:1])[0]
    assert np.allclose(hd[2:5], (0, 0, 1)) and np.allclose(hd[5:8], (0, 0, 1)) and np.allclose(hd[14:16], (0.25, 0.5))


def test_instance_transform_world_t(pkg, oracle_det):
    """Instance::hit (accel/mod.rs:122-138): scale 2 + translate; world t is re-derived from the world hit point."""
    xf = np.array([[2, 0, 0, 1], [0, 2, 0, 1], [0, 0, 2, -5], [0, 0, 0, 1]], dtype=np.float32)
    osc = oracle_det.scene(single_triangle_scene(pkg, xf))
    rays = np.zeros(1, dtype=pkg._abi.ray_dtype())
    rays["origin"], rays["direction"], rays["t_max"] = (1.5, 2.0, 1.0), (0, 0, -2.0), np.inf   # unnormalised direction
    h = osc.trace_closest(rays)[0]
    assert h["inst_idx"] == 0 and h["t"] == pytest.approx(3.0) and h["u"] == pytest.approx(0.25) and h["v"] == pytest.approx(0.5)
    hd = osc.trace_hitdata(rays)[0]
    assert np.allclose(hd[2:5], (0, 0, 1), atol=1e-6)   # (M^-1)^T n, renormalised


def test_closest_of_two_instances_and_tmax(pkg, oracle_det):
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material())
    m = b.add_mesh([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[0, 1, 2]])
    for z in (-1.0, -3.0, -2.0):
        t = np.eye(4, dtype=np.float32); t[2, 3] = z
        b.add_instance(m, t)
    b.set_sky_solid(1, 1, 1)
    osc = oracle_det.scene(b.build())
    rays = np.zeros(3, dtype=pkg._abi.ray_dtype())
    rays["origin"], rays["direction"] = (0, 0, 0), (0, 0, -1)
    rays["t_max"] = [np.inf, 0.5, 1.5]
    h = osc.trace_closest(rays)
    assert h["inst_idx"].tolist() == [0, pkg._abi.PTX_MISS, 0] and h["t"][0] == 1.0
    # hit_any tests the TLAS boxes against t_max, but instances restart with t_max = inf (accel/mod.rs:141-146)
    assert osc.trace_any(rays).tolist() == [1, 0, 1]


def test_white_furnace(pkg, oracle_libm):
    """Energy check: a white diffuse sphere under a constant unit sky.  The BRDF of brdf.rs is not exactly
    energy conserving (Schlick-weighted diffuse + multiscatter factor), so the band is loose: mean radiance
    over sphere pixels within [0.85, 1.05] of the sky."""
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material(base_color_factor=(1, 1, 1, 1), roughness_factor=1.0))
    n = 24
    th = (np.arange(n + 1) / n * 0.998 + 0.001) * np.pi
    ph = np.arange(2 * n + 1) / (2 * n) * 2 * np.pi
    P = np.array([[np.sin(t) * np.cos(-p), -np.cos(t), np.sin(t) * np.sin(-p)] for t in th for p in ph], dtype=np.float32)
    idx = []
    for j in range(n):
        for i in range(2 * n):
            a = j * (2 * n + 1) + i
            idx += [[a, a + 1, a + 2 * n + 2], [a, a + 2 * n + 2, a + 2 * n + 1]]
    b.add_instance(b.add_mesh(P, idx, normals=P))
    b.set_sky_solid(1, 1, 1)
    sc = b.build()
    osc = oracle_libm.scene(sc)
    cam = pkg.Camera.look_at((0, 0, 3), (0, 0, 0), 40.0, 1.0)
    w = h = 48
    acc = np.zeros((h, w, 3), dtype=np.float64)
    spp = 48
    for s in range(spp):
        acc += osc.render_pass(cam, w, h, s, 16)
    img = acc / spp
    hit_count = np.zeros((h, w), dtype=np.int32)
    for s in range(spp):   # the jitter moves silhouette pixels on and off the sphere
        hit_count += (osc.trace_closest(osc.primary_rays(cam, w, h, s))["inst_idx"] != pkg._abi.PTX_MISS).reshape(h, w)
    on, off = hit_count == spp, hit_count == 0
    assert on.sum() > 300
    assert np.allclose(img[off], 1.0, atol=1e-5)
    m = img[on].mean()
    assert 0.85 < m < 1.05, m
