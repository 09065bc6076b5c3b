"""Render-loop mirror over the HIP render core (ctypes over libmi355_pt.so).

`HipRenderLoop` has the shape of the reference's `RenderLoop` trait
(renderloop/mod.rs:9-22) and of `ParallelRenderLoop`'s inherent methods
(renderloop/parallel.rs:58-168): same names, same argument meaning, same
results layout (row-major Vec<Vec3> -> (H, W, 3) float32).  There is no CPU
fallback: if the HIP library or a GPU is missing every call raises.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
DEVICE_LIB_PATH = os.path.join(_HERE, "lib", "libmi355_pt.so")

EXPORTS = [
    "ptx_create", "ptx_destroy", "ptx_last_error", "ptx_upload_scene", "ptx_set_camera", "ptx_set_partition",
    "ptx_render_pass", "ptx_render_first_pass_aov", "ptx_render_pixels", "ptx_render_pass_unshaded",
    "ptx_fb_reset", "ptx_fb_render_passes", "ptx_fb_adaptive_pass", "ptx_fb_stats", "ptx_fb_read",
    "ptx_fb_read_mean", "ptx_fb_export_dev", "ptx_sync", "ptx_trace_batch", "ptx_generate_primary_rays",
    "ptx_set_counting", "ptx_reset_counters", "ptx_get_counters", "ptx_set_profiling", "ptx_reset_timings",
    "ptx_get_timings", "ptx_debug_math", "ptx_render_frame", "ptx_fb_tonemap_srgb8", "ptx_fb_adaptive_passes",
    "ptx_get_debug_stats", "ptx_trace_batch_noalloc", "ptx_fb_tiles_bytes", "ptx_fb_export_tiles_dev", "ptx_fb_import_tiles_dev",
    "ptx_order_after_stream", "ptx_order_stream_after", "ptx_fb_render_passes_async", "ptx_order_after_frame_tail",
    "ptx_fb_noisy_pairs", "ptx_fb_adaptive_pass_pixels",
]


class RenderCoreError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmi355_pt.so (no GPU needed to load it; needed for ptx_create)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(DEVICE_LIB_PATH):
        raise RenderCoreError(f"{DEVICE_LIB_PATH} is missing — build it with __graft_entry__.build(); "
                              "there is no CPU fallback for the render core")
    L = C.CDLL(DEVICE_LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.ptx_create.argtypes = [i32, C.POINTER(vp)]
    L.ptx_destroy.argtypes = [vp]
    L.ptx_last_error.argtypes = [vp]
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_upload_scene.argtypes = [vp, C.POINTER(abi.PtxSceneDesc)]
    L.ptx_set_camera.argtypes = [vp, C.POINTER(abi.PtxCamera)]
    L.ptx_set_partition.argtypes = [vp, u32, u32]
    L.ptx_render_pass.argtypes = [vp, u32, u32, u32, u32, vp]
    L.ptx_render_first_pass_aov.argtypes = [vp, u32, u32, u32, vp, vp, vp]
    L.ptx_render_pixels.argtypes = [vp, u32, u32, vp, u64, u32, vp]
    L.ptx_render_pass_unshaded.argtypes = [vp, u32, u32, u32, vp]
    L.ptx_fb_reset.argtypes = [vp, u32, u32, i32]
    L.ptx_fb_render_passes.argtypes = [vp, u32, u32, u32]
    L.ptx_fb_render_passes_async.argtypes = [vp, u32, u32, u32]
    L.ptx_fb_adaptive_pass.argtypes = [vp, C.POINTER(abi.PtxAdaptiveConfig), u32, C.POINTER(u64)]
    L.ptx_fb_adaptive_passes.argtypes = [vp, C.POINTER(abi.PtxAdaptiveConfig), u32, u32, C.POINTER(u32), C.POINTER(u64)]
    L.ptx_fb_noisy_pairs.argtypes = [vp, C.POINTER(abi.PtxAdaptiveConfig), vp, vp, u64, C.POINTER(u64)]
    L.ptx_fb_adaptive_pass_pixels.argtypes = [vp, vp, u64, u32, C.POINTER(u64)]
    L.ptx_fb_stats.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
    L.ptx_fb_read.argtypes = [vp, vp, vp, vp]
    L.ptx_fb_read_mean.argtypes = [vp, vp]
    L.ptx_fb_export_dev.argtypes = [vp, vp, vp, vp]
    L.ptx_sync.argtypes = [vp]
    L.ptx_fb_tonemap_srgb8.argtypes = [vp, i32, C.c_float, vp]
    L.ptx_render_frame.argtypes = [vp, u32, u32, u32, u32, C.POINTER(abi.PtxAdaptiveConfig), u32, vp, vp, C.POINTER(abi.PtxFrameStats)]
    L.ptx_trace_batch.argtypes = [vp, vp, u64, i32, vp, vp]
    L.ptx_generate_primary_rays.argtypes = [vp, u32, u32, u32, vp]
    L.ptx_set_counting.argtypes = [vp, i32]
    L.ptx_reset_counters.argtypes = [vp]
    L.ptx_get_counters.argtypes = [vp, C.POINTER(abi.PtxCounters)]
    L.ptx_set_profiling.argtypes = [vp, i32]
    L.ptx_reset_timings.argtypes = [vp]
    L.ptx_get_timings.argtypes = [vp, C.POINTER(abi.PtxTimings)]
    L.ptx_debug_math.argtypes = [vp, i32, vp, vp, u64, vp]
    L.ptx_get_debug_stats.argtypes = [vp, C.POINTER(abi.PtxDebugStats)]
    L.ptx_trace_batch_noalloc.argtypes = [vp, vp, u64, i32, vp, vp]
    L.ptx_fb_tiles_bytes.argtypes = [u32, u32, u32]
    L.ptx_fb_export_tiles_dev.argtypes = [vp, vp, u64]
    L.ptx_fb_import_tiles_dev.argtypes = [vp, vp, u32, u32]
    L.ptx_order_after_stream.argtypes = [vp, vp]
    L.ptx_order_after_frame_tail.argtypes = [vp, vp]
    L.ptx_order_stream_after.argtypes = [vp, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name == "ptx_fb_tiles_bytes":
            fn.restype = u64
        elif name != "ptx_last_error":
            fn.restype = i32
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class PathTracer:
    """`PathTracer<const MAX_BOUNCES: u32>` (renderer/pathtracer.rs:16): 8 in --benchmark, 16 in the app."""

    def __init__(self, max_bounces=8):
        self.max_bounces = int(max_bounces)


class UnshadedRenderer:
    """`UnshadedRenderer` (renderer/unshaded.rs:5): the moving-camera preview."""


class AdaptiveSamplingConfig:
    """`AdaptiveSamplingConfig` (renderloop/parallel.rs:13-17); app defaults: app.rs:139-143."""

    def __init__(self, min_samples=64, variance_threshold=0.001, max_samples_per_pass=None):
        self.min_samples = int(min_samples)
        self.variance_threshold = float(variance_threshold)
        self.max_samples_per_pass = max_samples_per_pass


class HipRenderLoop:
    """The render core behind the reference's `RenderLoop` interface.

    The reference borrows `&scene` / `&camera` on every call; here the scene
    is uploaded to HBM the first time it is seen (`Scene::new` happens once in
    the reference too) and the camera (80 bytes) is sent with every call.
    """

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        st = self.L.ptx_create(int(device), C.byref(h))
        if st != abi.PTX_OK:
            msg = self.L.ptx_last_error(None)
            raise RenderCoreError(f"ptx_create failed: {abi.ERROR_NAMES.get(st, st)}: {msg.decode() if msg else ''}")
        self._h = h
        self._scene = None

    # -- plumbing ---------------------------------------------------------------
    def _check(self, st, what):
        if st != abi.PTX_OK:
            msg = self.L.ptx_last_error(self._h)
            raise RenderCoreError(f"{what}: {abi.ERROR_NAMES.get(st, st)}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self.L.ptx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bind(self, scene, camera):
        if scene is not self._scene:
            self._check(self.L.ptx_upload_scene(self._h, scene.desc), "ptx_upload_scene")
            self._scene = scene
        if camera is not None:
            self._check(self.L.ptx_set_camera(self._h, C.byref(camera)), "ptx_set_camera")

    def set_partition(self, rank, world):
        self._check(self.L.ptx_set_partition(self._h, rank, world), "ptx_set_partition")

    def order_after_frame_tail(self, other):
        """Everything this loop enqueues from now on starts after `other`'s current frame is past its heavy iterations
        (`ptx_order_after_frame_tail`): two loops rendering alternate frames keep the GPU busy through each other's tails."""
        self._check(self.L.ptx_order_after_frame_tail(self._h, other._h), "ptx_order_after_frame_tail")

    def sync(self):
        self._check(self.L.ptx_sync(self._h), "ptx_sync")

    # -- RenderLoop::render_pass (renderloop/parallel.rs:171-194) -----------------
    def render_pass(self, scene, camera, renderer, width, height, sample_index):
        self._bind(scene, camera)
        out = np.zeros((height, width, 3), dtype=np.float32)
        if isinstance(renderer, UnshadedRenderer):
            self._check(self.L.ptx_render_pass_unshaded(self._h, width, height, sample_index, _p(out)), "ptx_render_pass_unshaded")
        else:
            self._check(self.L.ptx_render_pass(self._h, width, height, sample_index, renderer.max_bounces, _p(out)), "ptx_render_pass")
        return out

    # -- render_first_pass_with_aov (renderloop/parallel.rs:125-168) --------------
    def render_first_pass_with_aov(self, scene, camera, renderer, width, height):
        self._bind(scene, camera)
        out, alb, nrm = (np.zeros((height, width, 3), dtype=np.float32) for _ in range(3))
        self._check(self.L.ptx_render_first_pass_aov(self._h, width, height, renderer.max_bounces, _p(out), _p(alb), _p(nrm)),
                    "ptx_render_first_pass_aov")
        return out, (alb, nrm)

    # -- render_pixels (renderloop/parallel.rs:58-78) ------------------------------
    def render_pixels(self, scene, camera, renderer, width, height, pixels):
        self._bind(scene, camera)
        xys = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(-1, 3)
        out = np.zeros((len(xys), 3), dtype=np.float32)
        if len(xys):
            self._check(self.L.ptx_render_pixels(self._h, width, height, _p(xys), len(xys), renderer.max_bounces, _p(out)),
                        "ptx_render_pixels")
        return out

    # -- render_adaptive_pass (renderloop/parallel.rs:80-123) ----------------------
    def render_adaptive_pass(self, scene, camera, renderer, framebuffer, config):
        self._bind(scene, camera)
        framebuffer._require(self)
        mx = config.max_samples_per_pass
        if mx is None:
            mx = framebuffer.width * framebuffer.height
        cfg = abi.PtxAdaptiveConfig(config.min_samples, config.variance_threshold, mx)
        n = C.c_uint64(0)
        self._check(self.L.ptx_fb_adaptive_pass(self._h, C.byref(cfg), renderer.max_bounces, C.byref(n)), "ptx_fb_adaptive_pass")
        return int(n.value)

    def noisy_pairs(self, framebuffer, config):
        """`get_noisy_pixels` (framebuffer.rs:181-214) over the pixels this loop owns: (keys, pixels), two uint32 arrays.
        A larger key is a larger error (+inf below `min_samples`); pixel = y * width + x."""
        framebuffer._require(self)
        cfg = abi.PtxAdaptiveConfig(config.min_samples, config.variance_threshold, framebuffer.width * framebuffer.height)
        cap = framebuffer.width * framebuffer.height
        keys, pixels = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64(0)
        self._check(self.L.ptx_fb_noisy_pairs(self._h, C.byref(cfg), _p(keys), _p(pixels), cap, C.byref(n)), "ptx_fb_noisy_pairs")
        return keys[:n.value].copy(), pixels[:n.value].copy()

    def adaptive_pass_pixels(self, scene, camera, renderer, framebuffer, pixels):
        """One adaptive sample for each pixel of `pixels` (y * width + x, none twice), `sample_index` = the pixel's own
        count (parallel.rs:109-114), accumulated into the framebuffer.  Returns the number of samples added."""
        self._bind(scene, camera)
        framebuffer._require(self)
        pixels = np.ascontiguousarray(pixels, dtype=np.uint32)
        n = C.c_uint64(0)
        self._check(self.L.ptx_fb_adaptive_pass_pixels(self._h, _p(pixels), len(pixels), renderer.max_bounces, C.byref(n)), "ptx_fb_adaptive_pass_pixels")
        return int(n.value)

    def render_adaptive_passes(self, scene, camera, renderer, framebuffer, config, max_passes):
        """`render_adaptive_pass` up to `max_passes` times, stopping after the first pass that finds nothing noisy
        (app.rs:195-236) — same accumulators bit for bit, but several passes are traced together on the device
        (`ptx_fb_adaptive_passes`).  Returns (passes that rendered something, samples added)."""
        self._bind(scene, camera)
        framebuffer._require(self)
        mx = config.max_samples_per_pass
        if mx is None:
            mx = framebuffer.width * framebuffer.height
        cfg = abi.PtxAdaptiveConfig(config.min_samples, config.variance_threshold, mx)
        passes, n = C.c_uint32(0), C.c_uint64(0)
        self._check(self.L.ptx_fb_adaptive_passes(self._h, C.byref(cfg), renderer.max_bounces, max_passes, C.byref(passes), C.byref(n)),
                    "ptx_fb_adaptive_passes")
        return int(passes.value), int(n.value)

    # -- one whole frame: run_benchmark's `for sample` loop (main.rs:123-141) / app.rs:195-236 ----
    def render_frame(self, scene, camera, renderer, width, height, spp, adaptive=None, max_spp=0):
        """Returns (mean image h x w x 3, counts h x w, stats dict).  The framebuffer the call uses is the context's
        own; a DeviceFramebuffer bound to this loop is overwritten."""
        self._bind(scene, camera)
        mean = np.empty((height, width, 3), dtype=np.float32)
        counts = np.empty((height, width), dtype=np.uint32)
        fs = abi.PtxFrameStats()
        cfg = None
        if adaptive is not None:
            mx = adaptive.max_samples_per_pass
            cfg = abi.PtxAdaptiveConfig(adaptive.min_samples, adaptive.variance_threshold, width * height if mx is None else mx)
        self._check(self.L.ptx_render_frame(self._h, width, height, spp, renderer.max_bounces,
                                            C.byref(cfg) if cfg is not None else None, max_spp,
                                            _p(mean), _p(counts), C.byref(fs)), "ptx_render_frame")
        return mean, counts, {k: int(getattr(fs, k)) for k, _ in abi.PtxFrameStats._fields_}

    # -- Scene::hit / hit_any on a caller's batch (scene.rs:44-50) -----------------
    def trace_closest(self, scene, rays):
        self._bind(scene, None)
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype())
        out = np.zeros(len(rays), dtype=abi.hit_dtype())
        if len(rays):
            self._check(self.L.ptx_trace_batch(self._h, _p(rays), len(rays), abi.PTX_TRACE_CLOSEST, _p(out), None), "ptx_trace_batch")
        return out

    def trace_any(self, scene, rays):
        self._bind(scene, None)
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype())
        out = np.zeros(len(rays), dtype=np.uint8)
        if len(rays):
            self._check(self.L.ptx_trace_batch(self._h, _p(rays), len(rays), abi.PTX_TRACE_ANY, None, _p(out)), "ptx_trace_batch")
        return out

    def trace_batch_noalloc(self, scene, d_rays_ptr, n, mode, d_out_hits_ptr=None, d_out_any_ptr=None):
        """`ptx_trace_batch_noalloc`: rays and results are caller-owned device memory (raw pointers, e.g. torch
        `data_ptr()`); the call allocates nothing.  mode: abi.PTX_TRACE_CLOSEST / abi.PTX_TRACE_ANY."""
        self._bind(scene, None)
        self._check(self.L.ptx_trace_batch_noalloc(self._h, d_rays_ptr, int(n), int(mode), d_out_hits_ptr, d_out_any_ptr),
                    "ptx_trace_batch_noalloc")

    def primary_rays(self, scene, camera, width, height, sample_index):
        self._bind(scene, camera)
        out = np.zeros(width * height, dtype=abi.ray_dtype())
        self._check(self.L.ptx_generate_primary_rays(self._h, width, height, sample_index, _p(out)), "ptx_generate_primary_rays")
        return out

    def debug_math(self, fn, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, dtype=np.float32)
        out = np.zeros_like(x)
        code = {"sin": 0, "cos": 1, "exp": 2, "ln": 3, "atan2": 4, "asin": 5, "sincos_s": 6, "sincos_c": 7, "pow": 8, "log2": 9}[fn]
        self._check(self.L.ptx_debug_math(self._h, code, _p(x), _p(y), x.size, _p(out)), "ptx_debug_math")
        return out

    # -- measurement -----------------------------------------------------------------
    def set_counting(self, on):
        """False / 0 off; True / 1 the reference's events; 2 the shipped walk's events (include/ptx_core.h)."""
        self._check(self.L.ptx_set_counting(self._h, int(on)), "ptx_set_counting")

    def reset_counters(self):
        self._check(self.L.ptx_reset_counters(self._h), "ptx_reset_counters")

    def counters(self):
        c = abi.PtxCounters()
        self._check(self.L.ptx_get_counters(self._h, C.byref(c)), "ptx_get_counters")
        return c.as_dict()

    def debug_stats(self):
        """Stack high-water marks and packet-redo count since the last `reset_counters` (which traversal paths ran)."""
        d = abi.PtxDebugStats()
        self._check(self.L.ptx_get_debug_stats(self._h, C.byref(d)), "ptx_get_debug_stats")
        out = {}
        for n, _ in abi.PtxDebugStats._fields_:
            v = getattr(d, n)
            if n in ("_pad", "_pad2"):
                continue
            out[n] = float(v) if n.startswith("max_leaf_undershoot") else ([int(x) for x in v] if n.startswith("vote_") or n == "stack_ops" else int(v))
        return out

    def set_profiling(self, on):
        self._check(self.L.ptx_set_profiling(self._h, 1 if on else 0), "ptx_set_profiling")

    def reset_timings(self):
        self._check(self.L.ptx_reset_timings(self._h), "ptx_reset_timings")

    def timings(self):
        t = abi.PtxTimings()
        self._check(self.L.ptx_get_timings(self._h, C.byref(t)), "ptx_get_timings")
        return {n: getattr(t, n) for n, _ in abi.PtxTimings._fields_}


TILE = 8


def partition_pixels(width, height, rank, world):
    """Row-major pixel indices rank `rank` of `world` owns: 8x8 tiles, tile_index % world == rank, in the tile order
    the kernels use (one wave of primary rays == one tile).  Mirrors `ensure_worklist` in csrc/device/pt_core.hip."""
    tx_n, ty_n = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    t = np.arange(tx_n * ty_n, dtype=np.int64)
    t = t[t % world == rank]
    ly, lx = np.meshgrid(np.arange(TILE), np.arange(TILE), indexing="ij")
    x = (t % tx_n)[:, None] * TILE + lx.reshape(-1)[None, :]
    y = (t // tx_n)[:, None] * TILE + ly.reshape(-1)[None, :]
    ok = (x < width) & (y < height)
    return (y * width + x)[ok].astype(np.uint32)


def combine_framebuffers(dist, buffers):
    """The one exchange step of a multi-GPU frame: every rank holds full-size sum / sum_sq / count buffers that are
    zero outside its own tiles, so an all-reduce(SUM) (RCCL over xGMI on GPUs, gloo in the CPU tests) yields buffers
    bit-identical to the single-GPU ones (x + 0 == x; ownership is disjoint)."""
    for b in buffers:
        dist.all_reduce(b)
    return buffers


def tiles_bytes(width, height, world):
    """Bytes of one rank's packed tile block (`ptx_fb_tiles_bytes`): the same on every rank."""
    return int(lib().ptx_fb_tiles_bytes(width, height, world))


def pack_tiles(width, height, rank, world, sum_, sum_sq, counts):
    """Host mirror of `ptx_fb_export_tiles_dev` (numpy, for the CPU tests of the exchange): rank's tiles as
    (n_tiles_max * 64, 7) uint32 records."""
    tx_n, ty_n = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    n_max = -(-(tx_n * ty_n) // world)
    out = np.zeros((n_max * 64, 7), dtype=np.uint32)
    t = np.arange(rank, tx_n * ty_n, world, dtype=np.int64)
    ly, lx = np.meshgrid(np.arange(TILE), np.arange(TILE), indexing="ij")
    x = (t % tx_n)[:, None] * TILE + lx.reshape(-1)[None, :]
    y = (t // tx_n)[:, None] * TILE + ly.reshape(-1)[None, :]
    ok = (x < width) & (y < height)
    rec = np.nonzero(ok.reshape(-1))[0]
    p = (y * width + x)[ok]
    out[rec, 0:3] = np.ascontiguousarray(sum_, dtype=np.float32).reshape(-1, 3)[p].view(np.uint32)
    if sum_sq is not None:
        out[rec, 3:6] = np.ascontiguousarray(sum_sq, dtype=np.float32).reshape(-1, 3)[p].view(np.uint32)
    out[rec, 6] = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1)[p]
    return out


def unpack_tiles(width, height, src_rank, world, block, sum_, sum_sq, counts):
    """Host mirror of `ptx_fb_import_tiles_dev`: scatter `src_rank`'s block into full-size host buffers in place."""
    tx_n, ty_n = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    t = np.arange(src_rank, tx_n * ty_n, world, dtype=np.int64)
    ly, lx = np.meshgrid(np.arange(TILE), np.arange(TILE), indexing="ij")
    x = (t % tx_n)[:, None] * TILE + lx.reshape(-1)[None, :]
    y = (t // tx_n)[:, None] * TILE + ly.reshape(-1)[None, :]
    ok = (x < width) & (y < height)
    rec = np.nonzero(ok.reshape(-1))[0]
    p = (y * width + x)[ok]
    block = np.asarray(block, dtype=np.uint32).reshape(-1, 7)
    sum_.reshape(-1, 3)[p] = block[rec, 0:3].view(np.float32)
    if sum_sq is not None:
        sum_sq.reshape(-1, 3)[p] = block[rec, 3:6].view(np.float32)
    counts.reshape(-1)[p] = block[rec, 6]


class TileExchange:
    """The one collective of a multi-GPU frame: ONE all-gather of every rank's owned-tile records.

    `combine(fb)`: export this rank's tiles into a device buffer, all-gather the equal-sized blocks (RCCL over xGMI;
    `host_staged` = gloo rehearsal on a one-GPU box), scatter the other ranks' blocks into this rank's framebuffer.
    At 1080p the gathered buffer is 58 MB in total and every rank receives (world-1)/world of it — against three
    full-frame all-reduces (3 x 58 MB through a ring, 7/8 of it zeros at 8 ranks) before."""

    def __init__(self, dist, loop, width, height, rank, world, host_staged=False):
        import torch
        self.dist, self.loop, self.rank, self.world, self.host_staged = dist, loop, rank, world, host_staged
        self.nbytes = tiles_bytes(width, height, world)
        self.torch = torch
        self.send = torch.empty(self.nbytes, dtype=torch.uint8, device="cuda")
        self.recv = torch.empty(self.nbytes * world, dtype=torch.uint8, device="cuda")

    def combine(self, fb):
        """Export -> all-gather -> import, ordered by stream events only (`ptx_order_after_stream` /
        `ptx_order_stream_after`): the host never waits, so the caller can enqueue the next frame behind it."""
        torch, L, h = self.torch, self.loop.L, self.loop._h
        cur = torch.cuda.current_stream().cuda_stream      # hipStream_t of torch's current stream (0 = the default stream)
        # the previous frame's collective (enqueued behind torch's stream) must no longer read `send`
        self.loop._check(L.ptx_order_after_stream(h, cur), "ptx_order_after_stream")
        self.loop._check(L.ptx_fb_export_tiles_dev(h, self.send.data_ptr(), self.nbytes), "ptx_fb_export_tiles_dev")
        # the collective reads `send` after the export kernel on the library stream
        self.loop._check(L.ptx_order_stream_after(h, cur), "ptx_order_stream_after")
        if self.host_staged:
            parts = [torch.empty(self.nbytes, dtype=torch.uint8) for _ in range(self.world)]
            self.dist.all_gather(parts, self.send.cpu())          # .cpu() waits for torch's stream: gloo rehearsal only
            self.recv.copy_(torch.cat(parts))
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send)      # RCCL over xGMI; torch orders it on its current stream
        # the import kernels read `recv` after the collective
        self.loop._check(L.ptx_order_after_stream(h, cur), "ptx_order_after_stream")
        for r in range(self.world):
            if r != self.rank:
                self.loop._check(L.ptx_fb_import_tiles_dev(h, self.recv.data_ptr() + r * self.nbytes, r, self.world), "ptx_fb_import_tiles_dev")


def render_frame_distributed(all_reduce_max, loop, fb, scene, camera, renderer, spp, adaptive=None, max_spp=0):
    """One rank's share of `ptx_render_frame` when the frame is split over several GPUs (`loop.set_partition`).

    Uniform passes need no communication, and neither does the adaptive phase (app.rs:195-236): a pixel's sample
    sequence depends only on its own sums (it is sampled in a pass iff it is noisy then), every rank selects the noisy
    pixels among its own interleaved 8x8 tiles — that selection IS the variance-driven redistribution, the interleave
    spreads noisy regions over the ranks — and the global stop rule "no pixel beyond max_spp samples" is a cap of
    `max_spp - spp` passes on every rank, because all pixels start the adaptive phase with `spp` samples.  So each rank
    runs `render_adaptive_passes` on its own and the only collective is one 4-byte all-reduce(MAX) of the pass count
    for the frame statistics (`all_reduce_max(int) -> int`; torch.distributed over RCCL on GPUs, gloo or a stub in
    tests).  Afterwards `combine_framebuffers` merges the ranks' buffers; the result is bit-identical to the single-GPU
    frame.  Returns (this rank's adaptive sample count, adaptive passes of the whole frame)."""
    fb._require(loop)
    fb.reset()
    fb.render_passes(scene, camera, renderer, 0, spp)
    rendered, passes = 0, 0
    if adaptive is not None and max_spp > spp:
        passes, rendered = loop.render_adaptive_passes(scene, camera, renderer, fb, adaptive, max_spp - spp)
        passes = all_reduce_max(passes)
    return rendered, passes


def select_truncated(keys, pixels, max_samples_per_pass):
    """`render_adaptive_pass`'s cut (renderloop/parallel.rs:96-107) over the noisy pixels of ALL ranks: stable sort by
    error descending — `get_noisy_pixels` lists pixels in row-major order, so ties keep ascending pixel order — then
    `truncate(max)`.  `keys` / `pixels`: the concatenation of every rank's `noisy_pairs`.  Returns the surviving pixels
    in sorted order."""
    keys = np.asarray(keys, dtype=np.uint32)
    pixels = np.asarray(pixels, dtype=np.uint32)
    order = np.lexsort((pixels, -keys.astype(np.int64)))
    return pixels[order[:max(0, int(max_samples_per_pass))]]


def tile_owner(pixels, width, world):
    """Rank that owns each pixel under `ptx_set_partition` (8x8 tiles, tile index % world)."""
    pixels = np.asarray(pixels, dtype=np.int64)
    tx_n = (width + TILE - 1) // TILE
    return ((pixels // width // TILE) * tx_n + (pixels % width) // TILE) % world


def adaptive_pass_distributed(all_gather, loop, fb, scene, camera, renderer, config, rank, world):
    """One `render_adaptive_pass` (renderloop/parallel.rs:80-123) of a frame split over `world` GPUs, for configurations
    whose `max_samples_per_pass` may truncate the noisy list — the one case in which the ranks' lists have to meet,
    because the cut is taken over the error-sorted list of ALL noisy pixels.  Every rank lists its own noisy pixels
    (`ptx_fb_noisy_pairs`), ONE all-gather makes the lists global (`all_gather(keys, pixels) -> (all_keys, all_pixels)`:
    torch.distributed over RCCL or gloo, `gather_pairs_torch` below, or a stub in tests; at most 8 bytes per pixel of the
    frame), every rank takes the same cut (`select_truncated`) and samples the survivors among its own tiles
    (`ptx_fb_adaptive_pass_pixels`).  A pixel's sample depends only on its own count, so the ranks' framebuffers combine
    (`TileExchange`) to the single-GPU pass bit for bit.  Returns (samples this rank added, samples of the whole pass).
    Without truncation `render_frame_distributed` needs no exchange at all and traces 32+ passes together; use this only
    when `max_samples_per_pass` can bite."""
    mx = config.max_samples_per_pass
    if mx is None:
        mx = fb.width * fb.height
    keys, pixels = loop.noisy_pairs(fb, config)
    all_keys, all_pixels = all_gather(keys, pixels)
    chosen = select_truncated(all_keys, all_pixels, mx)
    mine = chosen[tile_owner(chosen, fb.width, world) == rank]
    added = loop.adaptive_pass_pixels(scene, camera, renderer, fb, mine) if len(mine) else 0
    return added, int(len(chosen))


def gather_pairs_torch(dist, device=None):
    """`all_gather` for `adaptive_pass_distributed` over torch.distributed: sizes first, then one padded all-gather of
    (key, pixel) pairs.  `device` = the tensors' device ("cuda" for the nccl backend = RCCL, None / "cpu" for gloo)."""
    import torch

    def gather(keys, pixels):
        world = dist.get_world_size()
        n = torch.tensor([len(keys)], dtype=torch.int64, device=device)
        sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        cap = max(1, max(sizes))
        mine = torch.zeros((cap, 2), dtype=torch.int64, device=device)
        if len(keys):
            mine[:len(keys), 0] = torch.from_numpy(keys.astype(np.int64)).to(mine.device)
            mine[:len(keys), 1] = torch.from_numpy(pixels.astype(np.int64)).to(mine.device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        k = np.concatenate([p[:s, 0].cpu().numpy() for p, s in zip(parts, sizes)]).astype(np.uint32)
        x = np.concatenate([p[:s, 1].cpu().numpy() for p, s in zip(parts, sizes)]).astype(np.uint32)
        return k, x
    return gather


class DeviceFramebuffer:
    """`Framebuffer` / `AdaptiveFramebuffer` (framebuffer.rs:10-60, 76-229) kept in HBM.

    One framebuffer per render loop (the context owns it): `accumulate` never
    sees host pixels — `render_passes` is `for sample { render_pass; accumulate }`
    (main.rs:124-133) entirely on the device.
    """

    def __init__(self, loop, width, height, adaptive=True):
        self.loop, self.width, self.height, self.adaptive = loop, int(width), int(height), bool(adaptive)
        self.reset()

    def _require(self, loop):
        if loop is not self.loop:
            raise RenderCoreError("framebuffer belongs to another render loop")

    def reset(self):
        self.loop._check(self.loop.L.ptx_fb_reset(self.loop._h, self.width, self.height, 1 if self.adaptive else 0), "ptx_fb_reset")

    def render_passes(self, scene, camera, renderer, first_sample, count, wait=True):
        """wait=False: `ptx_fb_render_passes_async` — returns once the work is enqueued; errors surface at the next
        synchronising call (`loop.sync()`, `stats()`, `read()`)."""
        self.loop._bind(scene, camera)
        if wait:
            self.loop._check(self.loop.L.ptx_fb_render_passes(self.loop._h, first_sample, count, renderer.max_bounces), "ptx_fb_render_passes")
        else:
            self.loop._check(self.loop.L.ptx_fb_render_passes_async(self.loop._h, first_sample, count, renderer.max_bounces), "ptx_fb_render_passes_async")

    def stats(self):
        mn, mx, tot = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        self.loop._check(self.loop.L.ptx_fb_stats(self.loop._h, C.byref(mn), C.byref(mx), C.byref(tot)), "ptx_fb_stats")
        return int(mn.value), int(mx.value), int(tot.value)

    def min_sample_count(self):
        return self.stats()[0]

    def max_sample_count(self):
        return self.stats()[1]

    def total_samples(self):
        return self.stats()[2]

    def read(self):
        s = np.zeros((self.height, self.width, 3), dtype=np.float32)
        q = np.zeros((self.height, self.width, 3), dtype=np.float32) if self.adaptive else None
        c = np.zeros((self.height, self.width), dtype=np.uint32)
        self.loop._check(self.loop.L.ptx_fb_read(self.loop._h, _p(s), _p(q), _p(c)), "ptx_fb_read")
        return s, q, c

    def to_color_buffer(self):
        m = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self.loop._check(self.loop.L.ptx_fb_read_mean(self.loop._h, _p(m)), "ptx_fb_read_mean")
        return m

    def to_srgb8(self, tonemap="aces", exposure=1.0):
        """`ToneMap::apply_with_exposure` + `linear_to_srgb_u8` (exporter/tonemapping.rs) on the device -> (H, W, 3) u8."""
        code = {"none": 0, "aces": 1, "reinhard": 2, "agx": 3}[tonemap]
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        self.loop._check(self.loop.L.ptx_fb_tonemap_srgb8(self.loop._h, code, float(exposure), _p(out)), "ptx_fb_tonemap_srgb8")
        return out

    def export_dev(self, d_sum_ptr, d_sum_sq_ptr, d_counts_ptr):
        """Copy the accumulators into caller-owned device memory (raw pointers, e.g. torch `data_ptr()`)."""
        self.loop._check(self.loop.L.ptx_fb_export_dev(self.loop._h, d_sum_ptr, d_sum_sq_ptr, d_counts_ptr), "ptx_fb_export_dev")
