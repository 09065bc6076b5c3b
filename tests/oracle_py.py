"""ctypes wrapper over oracle/lib/liboracle_{libm,det}.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  The product package never does.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _abi():
    import __graft_entry__ as ge
    return ge.load_package()._abi


class Oracle:
    """One build of the CPU restatement ("libm" = what a Rust build calls, "det" = bit-matches the GPU)."""

    def __init__(self, kind="det"):
        path = os.path.join(ROOT, "oracle", "lib", f"liboracle_{kind}.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        self.kind = kind
        abi = self.abi = _abi()
        L = self.L = C.CDLL(path)
        vp = C.c_void_p
        u32, u64, i32, f = C.c_uint32, C.c_uint64, C.c_int32, C.c_float
        L.orc_scene_new.argtypes = [C.POINTER(abi.PtxSceneDesc)]
        L.orc_scene_new.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_counters.argtypes = [C.POINTER(abi.PtxCounters)]
        L.orc_trace_batch.argtypes = [vp, vp, u64, i32, vp, vp]
        L.orc_trace_hitdata.argtypes = [vp, vp, u64, vp]
        L.orc_generate_primary_rays.argtypes = [vp, C.POINTER(abi.PtxCamera), u32, u32, u32, vp]
        L.orc_render_pass.argtypes = [vp, C.POINTER(abi.PtxCamera), u32, u32, u32, u32, vp]
        L.orc_render_first_pass_aov.argtypes = [vp, C.POINTER(abi.PtxCamera), u32, u32, u32, vp, vp, vp]
        L.orc_render_pass_unshaded.argtypes = [vp, C.POINTER(abi.PtxCamera), u32, u32, u32, vp]
        L.orc_render_pixels.argtypes = [vp, C.POINTER(abi.PtxCamera), u32, u32, vp, u64, u32, vp]
        L.orc_fb_new.argtypes = [u32, u32, i32]
        L.orc_fb_new.restype = vp
        L.orc_fb_free.argtypes = [vp]
        L.orc_fb_accumulate.argtypes = [vp, vp]
        L.orc_fb_render_passes.argtypes = [vp, vp, C.POINTER(abi.PtxCamera), u32, u32, u32]
        L.orc_fb_adaptive_pass.argtypes = [vp, vp, C.POINTER(abi.PtxCamera), C.POINTER(abi.PtxAdaptiveConfig), u32]
        L.orc_fb_adaptive_pass.restype = u64
        L.orc_fb_read.argtypes = [vp, vp, vp, vp]
        L.orc_fb_read_mean.argtypes = [vp, vp]
        L.orc_rng_stream.argtypes = [u32, u32, vp, vp]
        L.orc_bluenoise_sample.argtypes = [vp, C.c_int, u32, u32, u32, vp]
        L.orc_mis_weight.argtypes = [f, f]
        L.orc_mis_weight.restype = f
        L.orc_safe_inv_pdf.argtypes = [f]
        L.orc_safe_inv_pdf.restype = f
        L.orc_clamp_radiance.argtypes = [vp, vp]
        L.orc_evaluate_brdf.argtypes = [vp, vp, vp, f, vp, vp, vp]
        L.orc_fresnel_dielectric.argtypes = [f, f]
        L.orc_fresnel_dielectric.restype = f
        L.orc_sky_sample_direction.argtypes = [vp, f, f, vp]
        L.orc_sky_pdf.argtypes = [vp, vp]
        L.orc_sky_pdf.restype = f
        L.orc_sky_sample.argtypes = [vp, vp, vp]
        L.orc_texture_sample.argtypes = [vp, u32, f, f, vp]
        L.orc_math.argtypes = [C.c_int, vp, vp, u64, vp]
        L.orc_tonemap_srgb8.argtypes = [C.c_int, f, vp, u64, vp]
        L.orc_tonemap.argtypes = [C.c_int, f, vp, u64, vp]
        L.orc_trace_path.argtypes = [vp, vp, u32, u32, vp]
        assert L.orc_is_det_math() == (1 if kind == "det" else 0)

    def set_threads(self, n):
        self.L.orc_set_threads(int(n))

    def scene(self, host_scene):
        return OracleScene(self, host_scene)

    def reset_counters(self):
        self.L.orc_reset_counters()

    def counters(self):
        c = self.abi.PtxCounters()
        self.L.orc_get_counters(C.byref(c))
        return c.as_dict()

    def stack_overflowed(self):
        return bool(self.L.orc_stack_overflowed())

    # ---- unit hooks ----
    def rng_stream(self, seed, n):
        u = np.zeros(n, dtype=np.uint32)
        f = np.zeros(n, dtype=np.float32)
        self.L.orc_rng_stream(seed, n, _p(u), _p(f))
        return u, f

    def math(self, fn, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, dtype=np.float32)
        out = np.zeros_like(x)
        self.L.orc_math({"sin": 0, "cos": 1, "exp": 2, "ln": 3, "atan2": 4, "asin": 5, "pow": 6, "log2": 7}[fn], _p(x), _p(y), x.size, _p(out))
        return out

    def tonemap_srgb8(self, rgb, tonemap="aces", exposure=1.0):
        """ToneMap::apply_with_exposure + linear_to_srgb_u8 (exporter/tonemapping.rs) -> u8 array of rgb's shape."""
        a = np.ascontiguousarray(rgb, dtype=np.float32)
        out = np.zeros(a.shape, dtype=np.uint8)
        self.L.orc_tonemap_srgb8({"none": 0, "aces": 1, "reinhard": 2, "agx": 3}[tonemap], C.c_float(exposure), _p(a), a.size // 3, _p(out))
        return out

    def tonemap(self, rgb, tonemap="aces", exposure=1.0):
        a = np.ascontiguousarray(rgb, dtype=np.float32)
        out = np.zeros_like(a)
        self.L.orc_tonemap({"none": 0, "aces": 1, "reinhard": 2, "agx": 3}[tonemap], C.c_float(exposure), _p(a), a.size // 3, _p(out))
        return out

    def clamp_radiance(self, rgb):
        a = np.asarray(rgb, dtype=np.float32)
        o = np.zeros(3, dtype=np.float32)
        self.L.orc_clamp_radiance(_p(a), _p(o))
        return o

    def evaluate_brdf(self, v, l, n, alpha, f0, diffuse):
        arrs = [np.asarray(a, dtype=np.float32) for a in (v, l, n, f0, diffuse)]
        o = np.zeros(3, dtype=np.float32)
        self.L.orc_evaluate_brdf(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), alpha, _p(arrs[3]), _p(arrs[4]), _p(o))
        return o


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, oracle, host_scene):
        self.o = oracle
        self.host_scene = host_scene           # keeps the descriptor's arrays alive
        self._h = oracle.L.orc_scene_new(host_scene.desc)
        if not self._h:
            raise RuntimeError("orc_scene_new failed")

    def __del__(self):
        try:
            self.o.L.orc_scene_free(self._h)
        except Exception:
            pass

    def trace_closest(self, rays):
        abi = self.o.abi
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype())
        out = np.zeros(len(rays), dtype=abi.hit_dtype())
        self.o.L.orc_trace_batch(self._h, _p(rays), len(rays), abi.PTX_TRACE_CLOSEST, _p(out), None)
        return out

    def trace_any(self, rays):
        abi = self.o.abi
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype())
        out = np.zeros(len(rays), dtype=np.uint8)
        self.o.L.orc_trace_batch(self._h, _p(rays), len(rays), abi.PTX_TRACE_ANY, None, _p(out))
        return out

    def trace_hitdata(self, rays):
        rays = np.ascontiguousarray(rays, dtype=self.o.abi.ray_dtype())
        out = np.zeros((len(rays), 16), dtype=np.float32)
        self.o.L.orc_trace_hitdata(self._h, _p(rays), len(rays), _p(out))
        return out

    def primary_rays(self, cam, w, h, sample_index):
        out = np.zeros(w * h, dtype=self.o.abi.ray_dtype())
        self.o.L.orc_generate_primary_rays(self._h, C.byref(cam), w, h, sample_index, _p(out))
        return out

    def render_pass(self, cam, w, h, sample_index, max_bounces):
        out = np.zeros((h, w, 3), dtype=np.float32)
        self.o.L.orc_render_pass(self._h, C.byref(cam), w, h, sample_index, max_bounces, _p(out))
        return out

    def render_first_pass_with_aov(self, cam, w, h, max_bounces):
        out, alb, nrm = (np.zeros((h, w, 3), dtype=np.float32) for _ in range(3))
        self.o.L.orc_render_first_pass_aov(self._h, C.byref(cam), w, h, max_bounces, _p(out), _p(alb), _p(nrm))
        return out, alb, nrm

    def render_pass_unshaded(self, cam, w, h, sample_index=0):
        out = np.zeros((h, w, 3), dtype=np.float32)
        self.o.L.orc_render_pass_unshaded(self._h, C.byref(cam), w, h, sample_index, _p(out))
        return out

    def render_pixels(self, cam, w, h, xys, max_bounces):
        xys = np.ascontiguousarray(xys, dtype=np.uint32).reshape(-1, 3)
        out = np.zeros((len(xys), 3), dtype=np.float32)
        self.o.L.orc_render_pixels(self._h, C.byref(cam), w, h, _p(xys), len(xys), max_bounces, _p(out))
        return out

    def trace_path(self, origin, direction, seed, max_bounces):
        r = np.zeros(1, dtype=self.o.abi.ray_dtype())
        r["origin"][0] = origin
        r["direction"][0] = direction
        r["t_max"][0] = np.inf
        o = np.zeros(3, dtype=np.float32)
        self.o.L.orc_trace_path(self._h, _p(r), seed, max_bounces, _p(o))
        return o

    def sky_sample_direction(self, u1, u2):
        o = np.zeros(7, dtype=np.float32)
        self.o.L.orc_sky_sample_direction(self._h, u1, u2, _p(o))
        return o[:3].copy(), o[3:6].copy(), float(o[6])

    def sky_pdf(self, d):
        a = np.asarray(d, dtype=np.float32)
        return float(self.o.L.orc_sky_pdf(self._h, _p(a)))

    def sky_sample(self, d):
        a = np.asarray(d, dtype=np.float32)
        o = np.zeros(3, dtype=np.float32)
        self.o.L.orc_sky_sample(self._h, _p(a), _p(o))
        return o

    def texture_sample(self, tex, u, v):
        o = np.zeros(3, dtype=np.float32)
        self.o.L.orc_texture_sample(self._h, tex, u, v, _p(o))
        return o

    def bluenoise_sample(self, which, x, y, sample_index):
        o = np.zeros(2, dtype=np.float32)
        self.o.L.orc_bluenoise_sample(self._h, which, x, y, sample_index, _p(o))
        return o


class OracleFramebuffer:
    """`Framebuffer` / `AdaptiveFramebuffer` (framebuffer.rs) on the CPU."""

    def __init__(self, oscene, w, h, adaptive):
        self.s, self.w, self.h, self.adaptive = oscene, w, h, adaptive
        self._h = oscene.o.L.orc_fb_new(w, h, 1 if adaptive else 0)

    def __del__(self):
        try:
            self.s.o.L.orc_fb_free(self._h)
        except Exception:
            pass

    def render_passes(self, cam, first, count, max_bounces):
        self.s.o.L.orc_fb_render_passes(self._h, self.s._h, C.byref(cam), first, count, max_bounces)

    def adaptive_pass(self, cam, min_samples, threshold, max_per_pass, max_bounces):
        cfg = self.s.o.abi.PtxAdaptiveConfig(min_samples, threshold, max_per_pass)
        return int(self.s.o.L.orc_fb_adaptive_pass(self._h, self.s._h, C.byref(cam), C.byref(cfg), max_bounces))

    def read(self):
        s = np.zeros((self.h, self.w, 3), dtype=np.float32)
        q = np.zeros((self.h, self.w, 3), dtype=np.float32)
        c = np.zeros((self.h, self.w), dtype=np.uint32)
        self.s.o.L.orc_fb_read(self._h, _p(s), _p(q), _p(c))
        return s, q, c

    def mean(self):
        m = np.zeros((self.h, self.w, 3), dtype=np.float32)
        self.s.o.L.orc_fb_read_mean(self._h, _p(m))
        return m
