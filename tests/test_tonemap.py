"""Tone mapping + sRGB8 (exporter/tonemapping.rs:12-115): the oracle's restatement against hand-derived
values, and its `det` build (what the GPU evaluates) against its `libm` build (what a Rust build would compute)."""
import numpy as np
import pytest


def srgb_u8(c):
    c = np.clip(np.asarray(c, dtype=np.float64), 0.0, 1.0)
    s = np.where(c <= 0.0031308, 12.92 * c, 1.055 * c ** (1 / 2.4) - 0.055)
    return np.floor(s * 255.0).astype(np.int64)


def test_known_answers(oracle_libm):
    o = oracle_libm
    px = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.18, 0.18, 0.18], [4.0, 0.5, 0.02], [-1.0, 2.0, 0.001]], dtype=np.float32)
    # None: identity then the sRGB curve; 0 -> 0, negatives clamp to 0; white is 254, not 255: 1.055f32 - 0.055f32 is
    # 1 - 2^-24 and `as u8` truncates (linear_to_srgb_u8, tonemapping.rs:101-115) -- the reference's own rounding
    got = o.tonemap_srgb8(px, "none").astype(np.int64)
    assert np.abs(got - srgb_u8(px)).max() <= 1 and tuple(got[0]) == (0, 0, 0) and tuple(got[1]) == (254, 254, 254) and got[4, 0] == 0
    # Reinhard: c / (c + 1)
    assert np.allclose(o.tonemap(px[:4], "reinhard"), px[:4] / (px[:4] + 1.0), rtol=1e-6)
    # exposure multiplies first
    assert np.array_equal(o.tonemap(px, "reinhard", 2.0), o.tonemap(px * np.float32(2.0), "reinhard", 1.0))
    # ACES (Hill fit): in f64 with the same constants
    def aces(c):
        m1 = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
        m2 = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
        v = m1 @ c
        v = (v * (v + 0.0245786) - 0.000090537) / (v * (v * 0.983729 + 0.432951) + 0.238081)
        return np.clip(m2 @ v, 0.0, 1.0)
    want = np.stack([aces(p.astype(np.float64)) for p in px[:4]])
    assert np.allclose(o.tonemap(px[:4], "aces"), want, atol=2e-6)
    # AgX as written (its matrices are the transposes of the usual GLSL ones, so greys come out tinted): finite and
    # monotone in grey level from 1e-2 up; near black the contrast curve dips below 0 and powf(negative, 2.2) is NaN,
    # which `as u8` turns into 0
    greys = np.repeat(np.array([1e-2, 0.18, 1.0, 8.0], dtype=np.float32)[:, None], 3, axis=1)
    a = o.tonemap(greys, "agx")
    assert np.isfinite(a).all() and (np.diff(a[:, 1]) > 0).all() and 0.0 <= a[0, 1] < 0.01 and 0.8 < a[-1, 1] <= 1.05
    dark = np.full((1, 3), 1e-4, dtype=np.float32)
    assert np.isnan(o.tonemap(dark, "agx")).all() and (o.tonemap_srgb8(dark, "agx") == 0).all()


def test_det_pow_log2_are_within_one_ulp_of_glibc(oracle_det, oracle_libm):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0.0, 1.0, 20000), rng.uniform(1.0, 64.0, 5000), [0.0, 1.0, 0.5, 1e-10, 3.0e38]]).astype(np.float32)
    for y in (2.2, 1.0 / 2.4):
        yy = np.full_like(x, np.float32(y))
        d, l = oracle_det.math("pow", x, yy), oracle_libm.math("pow", x, yy)
        ulp = np.abs(d.view(np.int32).astype(np.int64) - l.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp == 0).mean() > 0.99
    xp = x[x > 0]
    d, l = oracle_det.math("log2", xp), oracle_libm.math("log2", xp)
    ulp = np.abs(d.view(np.int32).astype(np.int64) - l.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() > 0.99
    assert np.isnan(oracle_det.math("pow", np.array([-0.5], np.float32), np.array([2.2], np.float32))[0])


@pytest.mark.parametrize("mode,exposure", [("none", 1.0), ("aces", 1.5), ("reinhard", 1.0), ("agx", 1.0)])
def test_det_bytes_match_libm_bytes(oracle_det, oracle_libm, mode, exposure):
    rng = np.random.default_rng(11)
    px = np.exp(rng.uniform(np.log(1e-4), np.log(30.0), size=(40000, 3))).astype(np.float32)
    d = oracle_det.tonemap_srgb8(px, mode, exposure).astype(np.int64)
    l = oracle_libm.tonemap_srgb8(px, mode, exposure).astype(np.int64)
    assert np.abs(d - l).max() <= 1 and (d == l).mean() > 0.999
