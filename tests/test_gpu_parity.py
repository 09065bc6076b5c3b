"""GPU parity tests: the HIP render core, called through the C ABI, against the CPU oracle.

Bar (north_star): hit instance/triangle ids bit-exact; t/u/v bit-exact; fixed-seed images
bit-exact against the oracle's `det` build (same transcendental sequences as the kernels) and
within the stated tolerance against its `libm` build (what a Rust build would compute):
  relative difference of image means <= 1e-3 per channel (>= 16 spp, >= 4k pixels) and
  RMSE(gpu, cpu) <= 1.25 x RMSE(cpu, cpu') where cpu' = the oracle's own next-samples image.
"""
import ctypes as C

import numpy as np
import pytest

from oracle_py import OracleFramebuffer

pytestmark = pytest.mark.gpu

SCENES = [
    # name, flags, detail, w, h, bounces
    ("cornell", 0, 1.0, 96, 54, 8),
    ("cornell", 1, 1.0, 64, 36, 16),
    ("helmet_like", 0, 0.05, 96, 54, 8),
    ("sponza_like", 1, 0.05, 96, 54, 16),
    ("instanced_stress", 0, 0.05, 96, 54, 4),
    ("glass_test", 1, 0.3, 64, 36, 16),
    ("sss_test", 1, 0.3, 64, 36, 16),
    ("bistro_like", 3, 0.02, 96, 54, 16),
]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def scenes(pkg):
    cache = {}

    def get(name, flags, detail):
        key = (name, flags, detail)
        if key not in cache:
            cache[key] = pkg.Scene.procedural(name, 1, detail, flags)
        return cache[key]
    return get


def secondary_rays(abi, osc, rays, hits, seed):
    """Bounce rays (random hemisphere-ish directions from the hit points) and short segments (finite t_max)."""
    rng = np.random.default_rng(seed)
    ok = hits["inst_idx"] != abi.PTX_MISS
    o = rays["origin"][ok] + rays["direction"][ok] * hits["t"][ok, None]
    n = len(o)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    out = np.zeros(2 * n, dtype=abi.ray_dtype())
    out["origin"][:n] = o + d * np.float32(1e-3)
    out["direction"][:n] = d
    out["t_max"][:n] = np.inf
    out["origin"][n:] = o + d * np.float32(1e-3)
    out["direction"][n:] = -d
    out["t_max"][n:] = rng.uniform(0.01, 2.0, size=n).astype(np.float32)
    return out


@pytest.mark.parametrize("name,flags,detail,w,h,bounces", SCENES)
def test_hit_records_bit_exact(pkg, loop, oracle_det, scenes, name, flags, detail, w, h, bounces):
    abi = pkg._abi
    sc = scenes(name, flags, detail)
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    prim_c = osc.primary_rays(cam, w, h, 3)
    prim_g = loop.primary_rays(sc, cam, w, h, 3)
    assert np.array_equal(prim_c.view(np.uint8), prim_g.view(np.uint8)), "primary rays differ: RNG seeds would decorrelate (SURVEY F10)"
    hc = osc.trace_closest(prim_c)
    batches = [prim_c, secondary_rays(abi, osc, prim_c, hc, 7)]
    for rays in batches:
        c = osc.trace_closest(rays)
        g = loop.trace_closest(sc, rays)
        assert np.array_equal(c["inst_idx"], g["inst_idx"]), "instance ids differ"
        assert np.array_equal(c["tri_idx"], g["tri_idx"]), "triangle ids differ"
        for f in ("t", "u", "v"):
            assert np.array_equal(bits(c[f]), bits(g[f])), f"{f} bits differ"
        assert np.array_equal(osc.trace_any(rays), loop.trace_any(sc, rays)), "hit_any differs"
    assert not oracle_det.stack_overflowed()


@pytest.mark.parametrize("name,flags,detail,w,h,bounces", SCENES)
def test_packet_traversal_bit_exact(pkg, oracle_det, scenes, monkeypatch, name, flags, detail, w, h, bounces):
    """trace_packets (the wave's 64 rays walk the tree together; used for the primary rays of a frame) must return the
    per-ray traversal's hit records bit for bit — on coherent primaries, on incoherent bounce rays where the packet is
    the union of 64 unrelated walks, and on finite-t_max segments."""
    abi = pkg._abi
    monkeypatch.setenv("PTX_PACKET_BATCH", "1")
    lp = pkg.HipRenderLoop(0)                      # reads PTX_PACKET_BATCH at creation
    sc = scenes(name, flags, detail)
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    prim = osc.primary_rays(cam, w, h, 3)
    hc = osc.trace_closest(prim)
    for rays in (prim, secondary_rays(abi, osc, prim, hc, 7), prim[: 64 * 3 + 5]):     # the last one: a ragged final packet
        c = osc.trace_closest(rays)
        g = lp.trace_closest(sc, rays)
        assert np.array_equal(c.view(np.uint8), g.view(np.uint8))
        assert np.array_equal(osc.trace_any(rays), lp.trace_any(sc, rays)), "hit_any differs"      # any-hit packets


@pytest.mark.parametrize("name,flags,detail,w,h,bounces", [SCENES[3], SCENES[4], SCENES[7]])
def test_packet_stack_overflow_falls_back_to_per_ray(pkg, oracle_det, scenes, monkeypatch, name, flags, detail, w, h, bounces):
    """A packet's stack is the union of its rays' stacks; when it outgrows its capacity the packet's rays are redone by
    the per-ray kernel.  With the capacity cut to 10 entries most packets take that road: the image must not change."""
    monkeypatch.setenv("PTX_PACKET_CAPACITY", "10")
    lp = pkg.HipRenderLoop(0)                      # reads PTX_PACKET_CAPACITY at creation
    sc = scenes(name, flags, detail)
    cam = sc.default_camera(w / h)
    c = oracle_det.scene(sc).render_pass(cam, w, h, 1, bounces)
    g = lp.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, 1)
    assert np.array_equal(bits(c), bits(g))
    fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)          # a pixel-major batch of 8 passes
    fb.render_passes(sc, cam, pkg.PathTracer(bounces), 0, 8)
    ofb = OracleFramebuffer(oracle_det.scene(sc), w, h, True)
    ofb.render_passes(cam, 0, 8, bounces)
    assert np.array_equal(bits(ofb.read()[0]), bits(fb.read()[0]))


def test_axis_aligned_rays_nan_slabs(pkg, loop, oracle_det, scenes):
    """Rays with zero direction components whose origin lies on slab planes of axis-aligned walls:
    0 * inf = NaN inside the node test; the x86 min_ps/max_ps NaN rules decide the outcome."""
    abi = pkg._abi
    sc = scenes("cornell", 0, 1.0)
    osc = oracle_det.scene(sc)
    rays = []
    for ox in (-1.0, -0.3, 0.0, 0.3, 1.0):
        for oy in (0.0, 0.6, 1.2, 2.0):
            for oz in (-1.0, -0.3, 0.0, 0.35, 1.0, 2.5):
                for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (0, -1, 1), (-0.0, 0.0, -1.0)):
                    rays.append((ox, oy, oz, *d))
    a = np.zeros(len(rays), dtype=abi.ray_dtype())
    r = np.array(rays, dtype=np.float32)
    a["origin"], a["direction"], a["t_max"] = r[:, :3], r[:, 3:], np.inf
    c, g = osc.trace_closest(a), loop.trace_closest(sc, a)
    assert np.array_equal(c.view(np.uint8), g.view(np.uint8))
    assert np.array_equal(osc.trace_any(a), loop.trace_any(sc, a))


def test_traversal_counters_match_oracle(pkg, loop, oracle_det, scenes):
    """Traversal is bit-identical, so the event counts behind the algorithmic bytes must be too."""
    sc = scenes("sponza_like", 1, 0.05)
    cam = sc.default_camera(96 / 54)
    osc = oracle_det.scene(sc)
    rays = osc.primary_rays(cam, 96, 54, 0)
    for mode in ("closest", "any"):
        oracle_det.reset_counters()
        loop.set_counting(True)
        loop.reset_counters()
        if mode == "closest":
            osc.trace_closest(rays); loop.trace_closest(sc, rays)
        else:
            osc.trace_any(rays); loop.trace_any(sc, rays)
        loop.set_counting(False)
        c, g = oracle_det.counters(), loop.counters()
        for k in ("rays_closest", "rays_any", "node_visits", "node_children", "prim_index_reads", "triangle_tests",
                  "instance_entries", "instance_hits", "any_node_visits", "any_triangle_tests", "any_instance_entries"):
            assert c[k] == g[k], (mode, k, c[k], g[k])


@pytest.mark.parametrize("name,flags,detail,w,h,bounces", SCENES)
def test_render_pass_bit_exact_vs_det_oracle(pkg, loop, oracle_det, scenes, name, flags, detail, w, h, bounces):
    sc = scenes(name, flags, detail)
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    for sample_index in (0, 5):
        c = osc.render_pass(cam, w, h, sample_index, bounces)
        g = loop.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, sample_index)
        assert np.isfinite(g).all()
        assert np.array_equal(bits(c), bits(g)), f"{name}: {np.mean((bits(c) != bits(g)).any(axis=2)):.4%} of pixels differ"


def test_integrator_counters_match_oracle(pkg, loop, oracle_det, scenes):
    sc = scenes("sponza_like", 1, 0.05)
    cam = sc.default_camera(96 / 54)
    osc = oracle_det.scene(sc)
    oracle_det.reset_counters()
    osc.render_pass(cam, 96, 54, 1, 16)
    c = oracle_det.counters()
    loop.set_counting(True)
    loop.reset_counters()
    loop.render_pass(sc, cam, pkg.PathTracer(16), 96, 54, 1)
    loop.set_counting(False)
    g = loop.counters()
    for k in ("rays_closest", "rays_any", "node_visits", "node_children", "prim_index_reads", "triangle_tests", "instance_entries",
              "instance_hits", "material_fetches", "texture_lookups", "cdf_probes", "sky_pdf_evals", "sky_texel_lookups",
              "primary_rays", "pixel_samples_written"):
        assert c[k] == g[k], (k, c[k], g[k])
    # the kernels build hit data once per path vertex; the reference builds it once per instance that was hit
    assert g["hitdata_built"] <= c["hitdata_built"] and g["hitdata_built"] > 0


def test_aov_unshaded_and_render_pixels(pkg, loop, oracle_det, scenes):
    sc = scenes("helmet_like", 0, 0.05)
    w, h = 80, 45
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    c, ca, cn = osc.render_first_pass_with_aov(cam, w, h, 8)
    g, (ga, gn) = loop.render_first_pass_with_aov(sc, cam, pkg.PathTracer(8), w, h)
    assert np.array_equal(bits(c), bits(g)) and np.array_equal(bits(ca), bits(ga)) and np.array_equal(bits(cn), bits(gn))
    cu = osc.render_pass_unshaded(cam, w // 4, h // 4)
    gu = loop.render_pass(sc, cam, pkg.UnshadedRenderer(), w // 4, h // 4, 0)
    assert np.array_equal(bits(cu), bits(gu))
    rng = np.random.default_rng(3)
    work = np.stack([rng.integers(0, w, 500), rng.integers(0, h, 500), rng.integers(0, 300, 500)], axis=1).astype(np.uint32)
    work[10] = work[11]   # duplicates are legal in a work list
    cp = osc.render_pixels(cam, w, h, work, 8)
    gp = loop.render_pixels(sc, cam, pkg.PathTracer(8), w, h, work)
    assert np.array_equal(bits(cp), bits(gp))
    assert loop.render_pixels(sc, cam, pkg.PathTracer(8), w, h, np.zeros((0, 3), dtype=np.uint32)).shape == (0, 3)


def test_framebuffer_and_adaptive_passes(pkg, loop, oracle_det, scenes):
    """for sample { render_pass; accumulate } then render_adaptive_pass x3, all on the device (A1-A4)."""
    sc = scenes("cornell", 0, 1.0)
    w, h = 64, 36
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    ofb = OracleFramebuffer(osc, w, h, True)
    gfb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    ofb.render_passes(cam, 0, 6, 8)
    gfb.render_passes(sc, cam, pkg.PathTracer(8), 0, 6)
    cfg = pkg.AdaptiveSamplingConfig(min_samples=4, variance_threshold=0.02)
    for _ in range(3):
        nc = ofb.adaptive_pass(cam, 4, 0.02, w * h, 8)
        ng = loop.render_adaptive_pass(sc, cam, pkg.PathTracer(8), gfb, cfg)
        assert nc == ng and 0 < ng < w * h
    cs, cq, cc = ofb.read()
    gs, gq, gc = gfb.read()
    assert np.array_equal(cc, gc) and np.array_equal(bits(cs), bits(gs)) and np.array_equal(bits(cq), bits(gq))
    assert np.array_equal(bits(ofb.mean()), bits(gfb.to_color_buffer()))
    assert gfb.stats() == (int(cc.min()), int(cc.max()), int(cc.sum()))
    # truncating config: the stable descending sort by error decides which pixels get the samples
    cfg2 = pkg.AdaptiveSamplingConfig(min_samples=4, variance_threshold=0.0, max_samples_per_pass=100)
    nc = ofb.adaptive_pass(cam, 4, 0.0, 100, 8)
    ng = loop.render_adaptive_pass(sc, cam, pkg.PathTracer(8), gfb, cfg2)
    assert nc == ng == 100
    assert np.array_equal(ofb.read()[2], gfb.read()[2]) and np.array_equal(bits(ofb.read()[0]), bits(gfb.read()[0]))
    # below min_samples every pixel is selected with error +inf
    gfb.reset()
    gfb.render_passes(sc, cam, pkg.PathTracer(8), 0, 1)
    assert loop.render_adaptive_pass(sc, cam, pkg.PathTracer(8), gfb, pkg.AdaptiveSamplingConfig(64, 0.001)) == w * h


def test_render_frame_is_the_composed_loop(pkg, loop, oracle_det, scenes):
    """ptx_render_frame = fb_reset + spp uniform passes + adaptive passes until converged or capped, one readback."""
    sc = scenes("cornell", 1, 1.0)
    w, h = 64, 36
    cam = sc.default_camera(w / h)
    cfg = pkg.AdaptiveSamplingConfig(min_samples=4, variance_threshold=0.05)
    mean, counts, stats = loop.render_frame(sc, cam, pkg.PathTracer(8), w, h, 5, adaptive=cfg, max_spp=9)
    ofb = OracleFramebuffer(oracle_det.scene(sc), w, h, True)
    ofb.render_passes(cam, 0, 5, 8)
    passes = samples = 0
    while ofb.read()[2].max() < 9:
        n = ofb.adaptive_pass(cam, 4, 0.05, w * h, 8)
        if n == 0:
            break
        passes += 1
        samples += n
    assert passes >= 2
    assert stats["uniform_passes"] == 5 and stats["adaptive_passes"] == passes and stats["adaptive_samples"] == samples
    assert np.array_equal(counts, ofb.read()[2]) and np.array_equal(bits(mean), bits(ofb.mean()))
    assert stats["total_samples"] == int(counts.sum()) and stats["min_samples"] == int(counts.min()) and stats["max_samples"] == int(counts.max())
    # the benchmark's form: no adaptive phase, no variance buffer
    mean16, counts16, st16 = loop.render_frame(sc, cam, pkg.PathTracer(8), w, h, 16)
    plain = OracleFramebuffer(oracle_det.scene(sc), w, h, False)
    plain.render_passes(cam, 0, 16, 8)
    assert (counts16 == 16).all() and st16["adaptive_passes"] == 0 and np.array_equal(bits(mean16), bits(plain.mean()))


def test_pt_bench_benchmark_entry_point(pkg, loop, scenes):
    """The C++ host driver keeps the reference's `--benchmark` entry point and report (src/main.rs:77-152), and its
    frame is the one the Python mirror renders (same library, same scene, same camera)."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.render.DEVICE_LIB_PATH), "pt_bench")
    r = subprocess.run([exe, "--benchmark", "cornell"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    for line in ("Benchmark: cornell", "Loading model...", "Building acceleration structure...", "Rendering 16 samples at 640x360...",
                 "  Sample 4/16:", "  Sample 16/16:", "Results:", "  Build time:", "  Render time:", "  Primary rays: 3.69M",
                 "  Performance:", "Mprimary/s", "  Samples/sec:"):
        assert line in out, (line, out)
    sc = scenes("cornell", 0, 1.0)
    mean, counts, _ = loop.render_frame(sc, sc.default_camera(640 / 360), pkg.PathTracer(8), 640, 360, 16)
    got = float(re.search(r"Mean radiance: ([0-9.]+)", out).group(1))
    assert abs(got - float(mean.astype(np.float64).mean())) < 2e-6


def test_gltf_scene_renders_like_the_oracle(pkg, loop, oracle_det, tmp_path):
    """GLB -> load_gltf -> Scene::new -> render: textured, normal-mapped, transmissive materials from a file."""
    from gltf_fixture import write_test_glb
    path = str(tmp_path / "scene.glb")
    write_test_glb(path)
    g = pkg.gltf.load_gltf(path)
    sc = pkg.gltf.build_scene(g)
    w, h = 80, 45
    cam = pkg.gltf.camera_for(g, w / h)
    osc = oracle_det.scene(sc)
    for s in (0, 3):
        assert np.array_equal(bits(osc.render_pass(cam, w, h, s, 8)), bits(loop.render_pass(sc, cam, pkg.PathTracer(8), w, h, s)))


def test_device_tonemap_srgb8(pkg, loop, oracle_det, scenes, tmp_path):
    """ptx_fb_tonemap_srgb8: mean -> ToneMap::apply_with_exposure -> linear_to_srgb_u8 on the device, byte for byte
    the det oracle's (and pow/log2 bit for bit)."""
    sc = scenes("cornell", 1, 1.0)
    w, h = 64, 36
    cam = sc.default_camera(w / h)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=False)
    fb.render_passes(sc, cam, pkg.PathTracer(8), 0, 4)
    mean = fb.to_color_buffer()
    for mode, exposure in (("none", 1.0), ("aces", 1.5), ("reinhard", 0.7), ("agx", 1.0)):
        assert np.array_equal(fb.to_srgb8(mode, exposure), oracle_det.tonemap_srgb8(mean, mode, exposure)), mode
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0, 2, 5000), [0.0, 1.0, -1.0, np.inf]]).astype(np.float32)
    for y in (2.2, 1.0 / 2.4):
        yy = np.full_like(x, np.float32(y))
        assert np.array_equal(bits(loop.debug_math("pow", x, yy)), bits(oracle_det.math("pow", x, yy)))
    assert np.array_equal(bits(loop.debug_math("log2", x[:5002])), bits(oracle_det.math("log2", x[:5002])))
    # PngExporter (exporter/png.rs): the file holds exactly those bytes; the app exports AgX (app.rs:272)
    from PIL import Image
    path = str(tmp_path / "frame.png")
    pkg.PngExporter.with_tonemap("agx").with_exposure(1.0).export(fb, path)
    assert np.array_equal(np.asarray(Image.open(path)), oracle_det.tonemap_srgb8(mean, "agx", 1.0))
    pkg.PngExporter.srgb().export_adaptive(fb, path)
    assert np.array_equal(np.asarray(Image.open(path)), oracle_det.tonemap_srgb8(mean, "none", 1.0))
    with pytest.raises(NotImplementedError):
        pkg.PngExporter.srgb().with_denoiser(object())
    with pytest.raises(ValueError):
        pkg.PngExporter("filmic")


def test_truncating_adaptive_pass_split_over_ranks(pkg, loop, scenes, oracle_det):
    """`max_samples_per_pass` below the noisy count (renderloop/parallel.rs:96-107) on several GPUs: the cut is taken over
    the error-sorted list of ALL noisy pixels, so the ranks' lists meet in one all-gather (`adaptive_pass_distributed`:
    `ptx_fb_noisy_pairs` -> gather -> `select_truncated` -> `ptx_fb_adaptive_pass_pixels`).  Three virtual ranks = three
    contexts on this GPU, one thread each, a barrier standing in for the collective.  After every pass the combined
    buffers must equal the single-GPU `ptx_fb_adaptive_pass` AND the oracle's `render_adaptive_pass`, bit for bit."""
    import threading
    from oracle_py import OracleFramebuffer
    sc = scenes("cornell", 1, 1.0)
    w, h, spp, world, n_pass = 72, 40, 5, 3, 4
    cam = sc.default_camera(w / h)
    r8 = pkg.PathTracer(8)
    cfg = pkg.AdaptiveSamplingConfig(min_samples=4, variance_threshold=0.03, max_samples_per_pass=700)
    # single GPU and oracle, pass by pass
    single = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    single.render_passes(sc, cam, r8, 0, spp)
    ofb = OracleFramebuffer(oracle_det.scene(sc), w, h, True)
    ofb.render_passes(cam, 0, spp, 8)
    want, rendered = [], []
    for _ in range(n_pass):
        rendered.append(loop.render_adaptive_pass(sc, cam, r8, single, cfg))
        ofb.adaptive_pass(cam, cfg.min_samples, cfg.variance_threshold, cfg.max_samples_per_pass, 8)
        got, ora = single.read(), ofb.read()
        assert np.array_equal(bits(got[0]), bits(ora[0])) and np.array_equal(bits(got[1]), bits(ora[1])) and np.array_equal(got[2], ora[2])
        want.append(got)
    assert rendered[0] == 700 and len(loop.noisy_pairs(single, cfg)[0]) > 0          # the cut did bite
    barrier, pairs, parts, errors = threading.Barrier(world), [None] * world, [[None] * n_pass for _ in range(world)], []
    totals = [[0] * n_pass for _ in range(world)]

    def rank_main(r):
        try:
            def all_gather(keys, pixels):
                pairs[r] = (keys, pixels)
                barrier.wait()
                k = np.concatenate([p[0] for p in pairs]); x = np.concatenate([p[1] for p in pairs])
                barrier.wait()
                return k, x
            lp = pkg.HipRenderLoop(0)
            lp.set_partition(r, world)
            fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
            fb.render_passes(sc, cam, r8, 0, spp)
            for k in range(n_pass):
                _, totals[r][k] = pkg.render.adaptive_pass_distributed(all_gather, lp, fb, sc, cam, r8, cfg, r, world)
                parts[r][k] = fb.read()
            # a list with a pixel outside the framebuffer is refused, not traced
            with pytest.raises(pkg.RenderCoreError):
                lp.adaptive_pass_pixels(sc, cam, r8, fb, np.array([w * h], dtype=np.uint32))
        except Exception as e:   # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for k in range(n_pass):
        assert all(totals[r][k] == rendered[k] for r in range(world))
        total = [sum(parts[r][k][j] for r in range(world)) for j in range(3)]      # disjoint tiles + zeros
        assert np.array_equal(total[2], want[k][2])
        assert np.array_equal(bits(total[0]), bits(want[k][0])) and np.array_equal(bits(total[1]), bits(want[k][1]))


def test_adaptive_frame_split_over_ranks_is_the_single_gpu_frame(pkg, loop, scenes):
    """BASELINE config 5 on several GPUs: every rank runs uniform + adaptive passes on its interleaved tiles on its own
    (the pass cap is the same everywhere; one all-reduce(MAX) of the pass count for the statistics); the combined
    buffers equal the 1-GPU frame bit for bit.  Three virtual ranks = three contexts on this GPU, one thread each, a
    barrier standing in for the collective."""
    import threading
    sc = scenes("cornell", 1, 1.0)
    w, h, spp, cap, world = 72, 40, 5, 11, 3
    cam = sc.default_camera(w / h)
    cfg = pkg.AdaptiveSamplingConfig(min_samples=4, variance_threshold=0.03)
    mean, counts, stats = loop.render_frame(sc, cam, pkg.PathTracer(8), w, h, spp, adaptive=cfg, max_spp=cap)
    assert stats["adaptive_passes"] >= 3 and counts.max() == cap and counts.min() < cap
    barrier, vals, parts, errors = threading.Barrier(world), [0] * world, [None] * world, []

    def rank_main(r):
        try:
            def all_reduce_max(v):
                vals[r] = int(v)
                barrier.wait()
                m = max(vals)
                barrier.wait()
                return m
            lp = pkg.HipRenderLoop(0)
            lp.set_partition(r, world)
            fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
            _, passes = pkg.render.render_frame_distributed(all_reduce_max, lp, fb, sc, cam, pkg.PathTracer(8), spp, adaptive=cfg, max_spp=cap)
            assert passes == stats["adaptive_passes"]
            parts[r] = fb.read()
        except Exception as e:   # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    total = [sum(p[k] for p in parts) for k in range(3)]          # combine_framebuffers: disjoint tiles + zeros
    single = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)     # same frame again, raw accumulators this time
    pkg.render.render_frame_distributed(lambda v: int(v), loop, single, sc, cam, pkg.PathTracer(8), spp, adaptive=cfg, max_spp=cap)
    want = single.read()
    assert np.array_equal(want[2], counts) and np.array_equal(total[2], want[2])
    assert np.array_equal(bits(total[0]), bits(want[0])) and np.array_equal(bits(total[1]), bits(want[1]))


def _widen_bvh_in_place(bvh, rng):
    """Pull internal children's slots up into their parent while the parent stays <= 8 wide, and scatter the valid
    slots over the 8 positions: nodes a true 8-wide collapse would produce (the reference's own builder stops at 3)."""
    n = int(bvh.n_nodes)
    dt = np.dtype([("box", np.float32, (6, 8)), ("children", np.uint32, 8), ("valid", np.uint8), ("pad", np.uint8, 31)])
    assert dt.itemsize == 256
    nodes = np.ctypeslib.as_array(C.cast(bvh.nodes, C.POINTER(C.c_uint8)), shape=(n * 256,)).view(dt)
    widest = 0
    for i in range(n):
        slots = [(nodes[i]["box"][:, s].copy(), int(nodes[i]["children"][s])) for s in range(8) if nodes[i]["valid"] >> s & 1]
        grown = True
        while grown:
            grown = False
            for k, (_, word) in enumerate(slots):
                if word & 0x80000000 or word <= i:
                    continue
                sub = [(nodes[word]["box"][:, s].copy(), int(nodes[word]["children"][s])) for s in range(8) if nodes[word]["valid"] >> s & 1]
                if len(slots) - 1 + len(sub) <= 8:
                    slots[k:k + 1] = sub
                    grown = True
                    break
        pos = np.sort(rng.choice(8, size=len(slots), replace=False))
        nodes[i]["valid"] = 0
        nodes[i]["box"][:] = 0.0
        nodes[i]["children"][:] = 0
        for p, (box, word) in zip(pos, slots):
            nodes[i]["box"][:, p] = box
            nodes[i]["children"][p] = word
            nodes[i]["valid"] |= np.uint8(1 << int(p))
        widest = max(widest, len(slots))
    return widest


@pytest.mark.parametrize("name,detail,flags", [("sponza_like", 0.05, 1), ("instanced_stress", 0.3, 0)])
def test_nodes_wider_than_four_children(pkg, oracle_det, name, detail, flags):
    """Records hold 4 children; a 5-8 wide node becomes a chain whose pseudo child must keep the reference's slot
    order for leaves and for the LIFO of internal children.  Both sides traverse the same widened BVHs."""
    rng = np.random.default_rng(7)
    sc = pkg.Scene.procedural(name, 3, detail, flags)
    d = sc.desc.contents
    widest = _widen_bvh_in_place(d.tlas, rng)
    for m in range(d.n_meshes):
        widest = max(widest, _widen_bvh_in_place(d.meshes[m].bvh, rng))
    assert widest == 8
    w, h = 96, 54
    cam = sc.default_camera(w / h)
    loop = pkg.HipRenderLoop(0)
    osc = oracle_det.scene(sc)
    rays = loop.primary_rays(sc, cam, w, h, 0)
    hits = loop.trace_closest(sc, rays)
    ok = hits["inst_idx"] != pkg._abi.PTX_MISS
    assert ok.mean() > 0.1
    o = rays["origin"][ok] + rays["direction"][ok] * hits["t"][ok, None]
    dirs = rng.normal(size=o.shape).astype(np.float32)
    bounce = np.zeros(len(o), dtype=pkg._abi.ray_dtype())
    bounce["origin"], bounce["direction"], bounce["t_max"] = o, dirs, np.inf
    for batch in (rays, bounce):
        loop.set_counting(True); loop.reset_counters(); oracle_det.reset_counters()
        g = loop.trace_closest(sc, batch)
        c = osc.trace_closest(batch)
        gc, oc = loop.counters(), oracle_det.counters()
        loop.set_counting(False)
        for f in ("inst_idx", "tri_idx"):
            assert np.array_equal(g[f], c[f]), f
        for f in ("t", "u", "v"):
            assert np.array_equal(bits(g[f]), bits(c[f])), f
        for k in ("node_visits", "node_children", "triangle_tests", "instance_entries", "instance_hits", "prim_index_reads"):
            assert gc[k] == oc[k], (k, gc[k], oc[k])
        assert np.array_equal(loop.trace_any(sc, batch), osc.trace_any(batch))
    img_g = loop.render_pass(sc, cam, pkg.PathTracer(4), w, h, 1)
    img_c = osc.render_pass(cam, w, h, 1, 4)
    assert np.array_equal(bits(img_g), bits(img_c))


@pytest.mark.parametrize("name,flags,detail,batch", [("cornell", 1, 1.0, 3), ("sponza_like", 3, 0.05, 4), ("cornell", 0, 1.0, 16)])
def test_adaptive_passes_in_flight_equal_one_pass_at_a_time(pkg, scenes, monkeypatch, name, flags, detail, batch):
    """ptx_fb_adaptive_passes traces several adaptive passes together and replays them per pixel; sums, squares, counts,
    pass and sample totals must equal the render_adaptive_pass loop bit for bit (pixels that converge inside a batch
    drop their speculative samples)."""
    sc = scenes(name, flags, detail)
    w, h, spp, max_passes = 80, 45, 5, 7
    cam = sc.default_camera(w / h)
    cfg = pkg.AdaptiveSamplingConfig(min_samples=6, variance_threshold=0.04)
    r = pkg.PathTracer(8)
    monkeypatch.setenv("PTX_ADAPTIVE_BATCH", str(batch))
    lp = pkg.HipRenderLoop(0)                      # reads PTX_ADAPTIVE_BATCH at creation
    fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
    fb.render_passes(sc, cam, r, 0, spp)
    passes, samples = lp.render_adaptive_passes(sc, cam, r, fb, cfg, max_passes)
    got = fb.read()
    monkeypatch.setenv("PTX_ADAPTIVE_BATCH", "1")
    lp1 = pkg.HipRenderLoop(0)
    fb1 = pkg.DeviceFramebuffer(lp1, w, h, adaptive=True)
    fb1.render_passes(sc, cam, r, 0, spp)
    p1 = s1 = 0
    per_pass = []
    for _ in range(max_passes):
        n = lp1.render_adaptive_pass(sc, cam, r, fb1, cfg)
        if n == 0:
            break
        p1 += 1
        s1 += n
        per_pass.append(n)
    want = fb1.read()
    assert (passes, samples) == (p1, s1) and p1 >= 3
    assert per_pass[0] == w * h and per_pass[-1] < per_pass[1], per_pass      # pixels do converge inside the batches
    assert np.array_equal(got[2], want[2]) and np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(bits(got[1]), bits(want[1]))
    # counters on: one pass at a time, same result, and the counted primaries are the samples added
    fb.reset()
    fb.render_passes(sc, cam, r, 0, spp)
    lp.set_counting(True)
    lp.reset_counters()
    assert lp.render_adaptive_passes(sc, cam, r, fb, cfg, max_passes) == (p1, s1)
    assert lp.counters()["primary_rays"] == s1
    lp.set_counting(False)
    again = fb.read()
    assert np.array_equal(again[2], want[2]) and np.array_equal(bits(again[0]), bits(want[0]))


def test_passes_in_flight_do_not_change_the_sums(pkg, loop, scenes, monkeypatch):
    """ptx_fb_render_passes traces several passes per wavefront batch; the sums must equal one pass at a time (and the
    oracle's, test above) bit for bit, also when the last batch is short (7 passes as 3 + 3 + 1)."""
    sc = scenes("glass_test", 1, 1.0)
    w, h = 48, 40
    cam = sc.default_camera(w / h)
    ref = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    ref.render_passes(sc, cam, pkg.PathTracer(8), 2, 7)
    want = ref.read()
    for in_flight in (1, 3 * w * h):
        monkeypatch.setenv("PTX_PATHS_IN_FLIGHT", str(in_flight))
        other = pkg.HipRenderLoop(0)
        fb = pkg.DeviceFramebuffer(other, w, h, adaptive=True)
        fb.render_passes(sc, cam, pkg.PathTracer(8), 2, 7)
        got = fb.read()
        assert np.array_equal(want[2], got[2]) and np.array_equal(bits(want[0]), bits(got[0])) and np.array_equal(bits(want[1]), bits(got[1]))


def test_tile_partition_sums_to_whole_frame(pkg, loop, scenes):
    """Multi-GPU split (8x8 tiles, tile % world == rank): disjoint, and the sum over ranks is the 1-GPU frame, bit for bit."""
    sc = scenes("sponza_like", 1, 0.05)
    w, h = 100, 52   # not a multiple of 8 on purpose
    cam = sc.default_camera(w / h)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    fb.render_passes(sc, cam, pkg.PathTracer(8), 0, 3)
    whole = fb.read()
    world = 3
    acc = [np.zeros_like(whole[0]), np.zeros_like(whole[1]), np.zeros_like(whole[2])]
    try:
        for r in range(world):
            loop.set_partition(r, world)
            fb.reset()
            fb.render_passes(sc, cam, pkg.PathTracer(8), 0, 3)
            part = fb.read()
            assert (acc[2][part[2] > 0] == 0).all(), "tiles overlap"
            owned = np.flatnonzero(part[2].reshape(-1) > 0)
            assert np.array_equal(owned, np.sort(pkg.render.partition_pixels(w, h, r, world))), "python mirror of the partition differs"
            for a, p in zip(acc, part):
                a += p
    finally:
        loop.set_partition(0, 1)
    assert np.array_equal(acc[2], whole[2]) and np.array_equal(bits(acc[0]), bits(whole[0])) and np.array_equal(bits(acc[1]), bits(whole[1]))


def test_tile_exchange_kernels_match_host_mirror(pkg, loop, scenes):
    """ptx_fb_export_tiles_dev / ptx_fb_import_tiles_dev (the blocks one RCCL all-gather moves) against the numpy mirrors
    the 2-rank gloo test exchanges; three virtual ranks rebuild the whole frame in a fourth context, bit for bit."""
    from hip_ctypes import Hip
    sc = scenes("cornell", 1, 1.0)
    w, h, world = 100, 52, 3          # ragged edge tiles; 91 tiles: rank 0 owns one more than rank 2
    cam = sc.default_camera(w / h)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    fb.render_passes(sc, cam, pkg.PathTracer(8), 0, 3)
    whole = fb.read()
    nbytes = pkg.render.tiles_bytes(w, h, world)
    assert nbytes == -(-(13 * 7) // world) * 64 * 28
    hip = Hip()
    try:
        blocks = []
        for r in range(world):
            loop.set_partition(r, world)
            fb.reset()
            fb.render_passes(sc, cam, pkg.PathTracer(8), 0, 3)
            d = hip.malloc(nbytes)
            loop._check(loop.L.ptx_fb_export_tiles_dev(loop._h, d, nbytes), "export")
            loop.sync()
            got = hip.download(d, np.uint32, nbytes // 4).reshape(-1, 7)
            part = fb.read()
            assert np.array_equal(got, pkg.render.pack_tiles(w, h, r, world, *part)), f"rank {r}: export differs from the host mirror"
            blocks.append(d)
        loop.set_partition(0, 1)
        fb.reset()                       # an empty framebuffer receives every rank's block
        for r in range(world):
            loop._check(loop.L.ptx_fb_import_tiles_dev(loop._h, blocks[r], r, world), "import")
        got = fb.read()
        for a, b in zip(got, whole):
            assert np.array_equal(bits(a), bits(b))
        with pytest.raises(pkg.RenderCoreError):
            loop._check(loop.L.ptx_fb_export_tiles_dev(loop._h, blocks[0], nbytes - 1), "export")
    finally:
        loop.set_partition(0, 1)
        hip.release()


def test_image_within_tolerance_of_libm_oracle(pkg, loop, oracle_libm, scenes):
    """The reference-faithful comparison: the oracle with glibc transcendentals (what rustc's f32 methods call)."""
    for name, flags, detail, bounces in (("sponza_like", 1, 0.05, 16), ("sss_test", 1, 0.3, 16)):
        sc = scenes(name, flags, detail)
        w, h, spp = 96, 54, 32
        cam = sc.default_camera(w / h)
        osc = oracle_libm.scene(sc)
        a = OracleFramebuffer(osc, w, h, False); a.render_passes(cam, 0, spp, bounces)
        b = OracleFramebuffer(osc, w, h, False); b.render_passes(cam, spp, spp, bounces)
        g = pkg.DeviceFramebuffer(loop, w, h, adaptive=False)
        g.render_passes(sc, cam, pkg.PathTracer(bounces), 0, spp)
        ca, cb, gg = a.mean(), b.mean(), g.to_color_buffer()
        rel = np.abs(gg.mean(axis=(0, 1)) - ca.mean(axis=(0, 1))) / ca.mean(axis=(0, 1))
        assert (rel <= 1e-3).all(), (name, rel)
        rmse = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)))
        assert rmse(gg, ca) <= 1.25 * rmse(ca, cb), (name, rmse(gg, ca), rmse(ca, cb))
        # glibc's atan2f/asinf/sinf are 1 ulp off the correctly rounded value for 1-16 % of arguments (tests/test_oracle_math.py),
        # so few pixels are bit-identical; the gate is the tolerance above


def test_device_math_matches_oracle_det_and_glibc(pkg, loop, oracle_det, oracle_libm):
    rng = np.random.default_rng(11)
    n = 200000
    cases = {
        "sin": (rng.uniform(-7, 7, n), None), "cos": (rng.uniform(-7, 7, n), None),
        "exp": (rng.uniform(-100, 20, n), None), "ln": (10.0 ** rng.uniform(-12, 3, n), None),
        "atan2": (rng.normal(size=n), rng.normal(size=n)), "asin": (rng.uniform(-1, 1, n), None),
    }
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 3.1415927, 6.2831855, 88.7, -103.9, 1e-10], dtype=np.float32)
    for fn, (x, y) in cases.items():
        x = np.concatenate([x.astype(np.float32), edge])
        y = None if y is None else np.concatenate([y.astype(np.float32), edge[::-1]])
        g = loop.debug_math(fn, x, y)
        d = oracle_det.math(fn, x, y)
        m = oracle_libm.math(fn, x, y)
        both_nan = np.isnan(g) & np.isnan(d)
        assert (both_nan | (bits(g) == bits(d))).all(), f"{fn}: device and det oracle differ"
        fin = np.isfinite(m) & np.isfinite(g) & (np.abs(m) > 1e-30)
        ulp = np.abs(bits(g[fin]).astype(np.int64) - bits(m[fin]).astype(np.int64))
        assert ulp.max() <= 1, (fn, int(ulp.max()))
    x = rng.uniform(-7, 7, 1000).astype(np.float32)
    assert np.array_equal(bits(loop.debug_math("sincos_s", x)), bits(loop.debug_math("sin", x)))
    assert np.array_equal(bits(loop.debug_math("sincos_c", x)), bits(loop.debug_math("cos", x)))


def test_empty_and_error_paths(pkg, loop):
    abi = pkg._abi
    b = pkg.SceneBuilder()
    b.set_sky_solid(0.5, 0.6, 0.8)
    empty = b.build()   # Tlas::build with no instances (accel/mod.rs:167-173)
    cam = pkg.Camera.look_at((0, 1, 3), (0, 0.5, 0), 60.0, 16 / 9)
    rays = np.zeros(5, dtype=abi.ray_dtype())
    rays["direction"] = (0, 0, -1)
    rays["t_max"] = np.inf
    assert (loop.trace_closest(empty, rays)["inst_idx"] == abi.PTX_MISS).all()
    assert not loop.trace_any(empty, rays).any()
    img = loop.render_pass(empty, cam, pkg.PathTracer(8), 16, 9, 0)
    assert np.allclose(img, np.array([0.5, 0.6, 0.8], dtype=np.float32))
    assert len(loop.trace_closest(empty, rays[:0])) == 0
    with pytest.raises(pkg.RenderCoreError):
        loop.render_pixels(empty, cam, pkg.PathTracer(8), 16, 9, [[99, 0, 0]])
    with pytest.raises(pkg.RenderCoreError):
        loop.set_partition(2, 2)


def test_full_size_properties(pkg, loop, scenes):
    """BASELINE.json's sizes (1080p, PathTracer<16>): size-independent properties instead of an oracle render —
    run-to-run determinism, counts == spp, finite and clamped samples (clamp_radiance: max component <= 10),
    partition additivity, and an order-independent checksum."""
    sc = scenes("sponza_like", 1, 0.05)
    w, h, spp = 1920, 1080, 2
    cam = sc.default_camera(w / h)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    fb.render_passes(sc, cam, pkg.PathTracer(16), 0, spp)
    s1, q1, c1 = fb.read()
    fb.reset()
    fb.render_passes(sc, cam, pkg.PathTracer(16), 0, spp)
    s2, q2, c2 = fb.read()
    assert np.array_equal(bits(s1), bits(s2)) and np.array_equal(bits(q1), bits(q2)), "not deterministic run to run"
    assert (c1 == spp).all() and np.isfinite(s1).all() and s1.min() >= 0.0
    assert s1.max() <= 10.0 * spp * (1 + 1e-6)
    assert fb.stats() == (spp, spp, spp * w * h)
    try:
        loop.set_partition(1, 2)
        fb.reset()
        fb.render_passes(sc, cam, pkg.PathTracer(16), 0, spp)
        sp, _, cp = fb.read()
    finally:
        loop.set_partition(0, 1)
    own = cp > 0
    assert abs(own.mean() - 0.5) < 0.01 and np.array_equal(bits(sp[own]), bits(s1[own])) and (sp[~own] == 0).all()


def test_upload_rejects_malformed_acceleration_structures(pkg):
    """`ptx_upload_scene` validates what the kernels will chase: child words in range, leaf ranges in range, and the node
    graph a TREE (a cycle would never end on the GPU; the collapse at upload walks it too).  Nothing is launched."""
    lp = pkg.HipRenderLoop(0)
    dt = np.dtype([("box", np.float32, (6, 8)), ("children", np.uint32, 8), ("valid", np.uint8), ("pad", np.uint8, 31)])

    def nodes_of(bvh):
        return np.ctypeslib.as_array(C.cast(bvh.nodes, C.POINTER(C.c_uint8)), shape=(int(bvh.n_nodes) * 256,)).view(dt)

    def corrupt(mutate, what):
        sc = pkg.Scene.procedural("sponza_like", 1, 0.05, 0)
        mutate(sc.desc.contents)
        with pytest.raises(pkg.RenderCoreError, match="PTX_ERR_INVALID"):
            lp._bind(sc, None)
        assert lp._scene is None or lp._scene is not sc, what

    def internal_slot(nodes):
        for i in range(len(nodes)):
            for s in range(8):
                if nodes[i]["valid"] >> s & 1 and not nodes[i]["children"][s] & 0x80000000:
                    return i, s
        raise AssertionError("no internal child")

    def cycle(d):
        n = nodes_of(d.meshes[0].bvh)
        i, s = internal_slot(n)
        n[i]["children"][s] = 0                     # a child pointing back at the root
    corrupt(cycle, "cycle")

    def shared_subtree(d):
        n = nodes_of(d.tlas)
        i, s = internal_slot(n)
        j, t = next((a, b) for a in range(len(n)) for b in range(8)
                    if (a, b) != (i, s) and n[a]["valid"] >> b & 1 and not n[a]["children"][b] & 0x80000000)
        n[j]["children"][t] = n[i]["children"][s]   # two parents for one node: a DAG, not a tree
    corrupt(shared_subtree, "dag")

    def out_of_range_child(d):
        n = nodes_of(d.tlas)
        i, s = internal_slot(n)
        n[i]["children"][s] = int(d.tlas.n_nodes)
    corrupt(out_of_range_child, "child index")

    def leaf_past_the_end(d):
        n = nodes_of(d.meshes[0].bvh)
        i = next(a for a in range(len(n)) if any(n[a]["valid"] >> b & 1 and n[a]["children"][b] & 0x80000000 for b in range(8)))
        s = next(b for b in range(8) if n[i]["valid"] >> b & 1 and n[i]["children"][b] & 0x80000000)
        n[i]["children"][s] = 0x80000000 | (3 << 24) | (int(d.meshes[0].bvh.n_primitive_indices) - 1)
    corrupt(leaf_past_the_end, "leaf range")

    def empty_node(d):
        nodes_of(d.tlas)[0]["valid"] = 0
    corrupt(empty_node, "empty node")
    # and a good scene still uploads afterwards
    good = pkg.Scene.procedural("sponza_like", 1, 0.05, 0)
    lp._bind(good, None)
    assert lp._scene is good


def test_host_tree_with_loose_boxes_is_walked_as_given(pkg, oracle_det, monkeypatch):
    """The collapsed trees are only equivalent when an internal child's box contains its own children's boxes.  A host
    BVH that breaks this (here: every internal box of every mesh BVH shrunk by 30 %) keeps its faithful records for all
    kernels; hits must equal the oracle's walk of the same arrays, per ray and as packets."""
    sc = pkg.Scene.procedural("sponza_like", 2, 0.05, 1)
    d = sc.desc.contents
    dt = np.dtype([("box", np.float32, (6, 8)), ("children", np.uint32, 8), ("valid", np.uint8), ("pad", np.uint8, 31)])
    shrunk = 0
    for m in range(d.n_meshes):
        bvh = d.meshes[m].bvh
        nodes = np.ctypeslib.as_array(C.cast(bvh.nodes, C.POINTER(C.c_uint8)), shape=(int(bvh.n_nodes) * 256,)).view(dt)
        for i in range(len(nodes)):
            for s in range(8):
                if nodes[i]["valid"] >> s & 1 and not nodes[i]["children"][s] & 0x80000000:
                    b = nodes[i]["box"][:, s]           # min_x max_x min_y max_y min_z max_z
                    for a in range(3):
                        c, h = 0.5 * (b[2 * a] + b[2 * a + 1]), 0.35 * (b[2 * a + 1] - b[2 * a])
                        b[2 * a], b[2 * a + 1] = c - h, c + h
                    shrunk += 1
    assert shrunk > 100
    w, h = 160, 90
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    rays = osc.primary_rays(cam, w, h, 0)
    want, want_any = osc.trace_closest(rays), osc.trace_any(rays)
    lp = pkg.HipRenderLoop(0)
    monkeypatch.setenv("PTX_PACKET_BATCH", "1")
    lpk = pkg.HipRenderLoop(0)
    for loop_ in (lp, lpk):
        assert np.array_equal(loop_.trace_closest(sc, rays).view(np.uint8), want.view(np.uint8))
        assert np.array_equal(loop_.trace_any(sc, rays), want_any)
    assert np.array_equal(bits(lp.render_pass(sc, cam, pkg.PathTracer(8), w, h, 1)), bits(osc.render_pass(cam, w, h, 1, 8)))
    # the shrunk boxes do lose hits: this tree is not the one the builder made
    good = pkg.Scene.procedural("sponza_like", 2, 0.05, 1)
    assert not np.array_equal(oracle_det.scene(good).trace_closest(rays)["tri_idx"], want["tri_idx"])


def _uv_sphere(n_lat=10, n_lon=16):
    pos = [(0.0, 1.0, 0.0)]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            pos.append((np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)))
    pos.append((0.0, -1.0, 0.0))
    idx = []
    ring = lambda i, j: 1 + (i - 1) * n_lon + j % n_lon
    for j in range(n_lon):
        idx.append((0, ring(1, j), ring(1, j + 1)))
        idx.append((len(pos) - 1, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            idx.append((ring(i, j), ring(i + 1, j), ring(i + 1, j + 1)))
            idx.append((ring(i, j), ring(i + 1, j + 1), ring(i, j + 1)))
    return np.array(pos, dtype=np.float32), np.array(idx, dtype=np.uint32)


def _rot(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def _affine(lin, t):
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = lin, t
    return m


def _rays_at_instances(abi, pos, transforms, seed, per_instance=400):
    """Rays aimed at points of every instance's surface: infinite ones, segments that end within a few percent of the
    surface (either side: the band a forwarded bound must not cut into), and directions far from unit length."""
    rng = np.random.default_rng(seed)
    out = []
    for m in transforms:
        world = pos.astype(np.float64) @ m[:3, :3].T + m[:3, 3]
        size = np.linalg.norm(world.max(0) - world.min(0))
        target = world[rng.integers(0, len(world), per_instance)] * (1 - 0.2 * rng.uniform(size=(per_instance, 1))) + \
            0.2 * rng.uniform(size=(per_instance, 1)) * world.mean(0)
        origin = target + rng.normal(size=(per_instance, 3)) * size * rng.choice([0.6, 2.0, 8.0], size=(per_instance, 1))
        d = target - origin
        dist = np.linalg.norm(d, axis=1, keepdims=True)
        d /= dist
        scale = rng.choice([1.0, 1.0, 1.0, 0.1, 7.0], size=(per_instance, 1))          # |d| = 1, 0.1 or 7
        r = np.zeros(per_instance, dtype=abi.ray_dtype())
        r["origin"], r["direction"] = origin.astype(np.float32), (d * scale).astype(np.float32)
        tm = np.where(rng.uniform(size=per_instance) < 0.5, np.inf, (dist[:, 0] / scale[:, 0]) * rng.uniform(0.9, 1.1, per_instance))
        r["t_max"] = tm.astype(np.float32)
        out.append(r)
    return np.concatenate(out)


def test_forwarded_bound_with_hard_instance_transforms(pkg, loop, oracle_det, monkeypatch):
    """`forwarded_bound` (pt_traverse.h) rests on a per-scene certified bound for the world -> local -> world round
    trip.  Instances far from the origin, tiny, huge, anisotropic, sheared and thin — each with a twin whose surface
    nearly coincides (offsets from one part in 10^7 to one in 10^2 of the instance's size, so that world distances of
    different instances differ by roundings only) — must give the oracle's bytes per ray and as packets.  A scene
    whose instance matrices are too ill-conditioned for the bound switches forwarding off by itself."""
    abi = pkg._abi
    pos, idx = _uv_sphere()
    lins = [
        (np.eye(3) * 0.01, (5000.0, -3000.0, 2000.0)),
        (_rot((1, 2, 3), 1.1) @ np.diag([50.0, 1.0, 0.2]), (3.0, 1.0, -2.0)),
        (np.array([[1.0, 2.0, 0.0], [0.0, 1.0, 3.0], [0.0, 0.0, 1.0]]), (-40.0, 7.0, 11.0)),
        (np.eye(3) * 300.0, (100.0, 900.0, -250.0)),
        (_rot((0, 1, 1), 0.4) @ np.diag([1.0, 1.0, 1e-3]), (0.0, 0.0, 0.0)),
        (_rot((2, -1, 0.5), 2.3) @ np.diag([4.0, 0.5, 2.0]) @ _rot((0, 0, 1), 0.7), (12.0, -6.0, 30.0)),
    ]
    transforms = []
    rng = np.random.default_rng(5)
    for k, (lin, t) in enumerate(lins):
        m = _affine(lin, t)
        transforms.append(m)
        size = np.linalg.norm(lin, 2)
        for rel in (1e-7, 1e-4, 1e-2):                       # twins: nearly coincident surfaces in other instances
            transforms.append(_affine(lin, np.asarray(t) + rng.normal(size=3) * size * rel))

    def build(ts):
        b = pkg.SceneBuilder()
        b.add_material(pkg.default_material())
        mesh = b.add_mesh(pos, idx)
        order = rng.permutation(len(ts))                     # TLAS order unrelated to distance
        for i in order:
            b.add_instance(mesh, ts[i].astype(np.float32))
        b.set_sky_solid(0.5, 0.6, 0.8)
        return b.build()

    sc = build(transforms)
    osc = oracle_det.scene(sc)
    rays = _rays_at_instances(abi, pos, transforms, 17)
    want, want_any = osc.trace_closest(rays), osc.trace_any(rays)
    assert (want["inst_idx"] != abi.PTX_MISS).mean() > 0.2
    monkeypatch.setenv("PTX_PACKET_BATCH", "1")
    lpk = pkg.HipRenderLoop(0)
    monkeypatch.delenv("PTX_PACKET_BATCH")
    for loop_ in (loop, lpk):
        got = loop_.trace_closest(sc, rays)
        assert loop_.debug_stats()["bound_forwarding"] == 1
        bad = np.flatnonzero((got.view(np.uint8).reshape(len(rays), -1) != want.view(np.uint8).reshape(len(rays), -1)).any(axis=1))
        assert len(bad) == 0, (len(bad), rays[bad[:3]], got[bad[:3]], want[bad[:3]])
        assert np.array_equal(loop_.trace_any(sc, rays), want_any)
    # coherent bundles through the twins (what the packet kernel is for)
    cam = pkg.Camera.look_at((3.0, 1.0, 60.0), (3.0, 1.0, -2.0), 50.0, 16 / 9)
    prim = osc.primary_rays(cam, 160, 90, 0)
    for loop_ in (loop, lpk):
        assert np.array_equal(loop_.trace_closest(sc, prim).view(np.uint8), osc.trace_closest(prim).view(np.uint8))
    # an instance squeezed by 10^5 between two rotations: |M||M^-1| is ~10^5, the bound would be useless -> forwarding off
    ill = transforms[:4] + [_affine(_rot((1, 1, 0), 0.8) @ np.diag([1.0, 1.0, 1e-5]) @ _rot((0, 1, 2), 1.3), (1.0, 2.0, 3.0))]
    sc2 = build(ill)
    osc2 = oracle_det.scene(sc2)
    rays2 = _rays_at_instances(abi, pos, ill, 23, 200)
    got2 = loop.trace_closest(sc2, rays2)
    assert loop.debug_stats()["bound_forwarding"] == 0
    assert np.array_equal(got2.view(np.uint8), osc2.trace_closest(rays2).view(np.uint8))


def test_nearest_first_walk_with_near_coincident_surfaces(pkg, loop, oracle_det, monkeypatch):
    """The closest-hit kernels walk nearest child first and hand a ray back to the reference order only when a second
    hit lies within a small band of the best one (csrc/device/pt_traverse.h "FAST mode").  Here most rays have such
    a second hit: one mesh holds stacked copies of a grid — exact duplicates (ties the reference settles by its walk
    order), copies 1e-6, 3e-4, 2e-3 and 5e-2 above — with its triangles shuffled, instanced three times so that
    surfaces of different instances nearly coincide as well.  Every hit must be the oracle's, and the kernel must
    report that it took the second road."""
    abi = pkg._abi
    rng = np.random.default_rng(11)
    n = 9
    gx, gz = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    layers = [0.0, 0.0, 1e-6, 3e-4, 2e-3, 5e-2]
    pos, idx = [], []
    for y in layers:
        base = len(pos)
        pos += [(float(x), y, float(z)) for x, z in zip(gx.ravel(), gz.ravel())]
        for i in range(n - 1):
            for j in range(n - 1):
                a, b, c, d = base + i * n + j, base + (i + 1) * n + j, base + (i + 1) * n + j + 1, base + i * n + j + 1
                idx += [(a, b, c), (a, c, d)]
    idx = np.array(idx, dtype=np.uint32)[rng.permutation(len(idx))]
    pos = np.array(pos, dtype=np.float32)
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material())
    mesh = b.add_mesh(pos, idx)
    transforms = [np.eye(4), _affine(np.eye(3), (0.0, 1e-4, 0.0)), _affine(_rot((0, 1, 0), 0.5) * 1.5, (0.1, -2e-4, 0.0)),
                  _affine(_rot((1, 0, 0), 0.3), (0.0, 0.02, 0.3))]
    for t in transforms:
        b.add_instance(mesh, np.asarray(t, dtype=np.float32))
    b.set_sky_solid(0.5, 0.6, 0.8)
    sc = b.build()
    osc = oracle_det.scene(sc)
    m = 60000
    target = np.stack([rng.uniform(-1.2, 1.2, m), rng.uniform(-0.01, 0.06, m), rng.uniform(-1.2, 1.2, m)], axis=1)
    origin = target + rng.normal(size=(m, 3)) * rng.choice([0.05, 0.5, 3.0], size=(m, 1))
    d = target - origin
    dist = np.linalg.norm(d, axis=1, keepdims=True)
    d /= dist
    rays = np.zeros(m, dtype=abi.ray_dtype())
    rays["origin"], rays["direction"] = origin.astype(np.float32), d.astype(np.float32)
    rays["t_max"] = np.where(rng.uniform(size=m) < 0.6, np.inf, dist[:, 0] * rng.uniform(0.97, 1.03, m)).astype(np.float32)
    want = osc.trace_closest(rays)
    assert (want["inst_idx"] != abi.PTX_MISS).mean() > 0.3
    loop.reset_counters()
    got = loop.trace_closest(sc, rays)
    dbg = loop.debug_stats()
    bad = np.flatnonzero((got.view(np.uint8).reshape(m, -1) != want.view(np.uint8).reshape(m, -1)).any(axis=1))
    assert len(bad) == 0, (len(bad), rays[bad[:3]], got[bad[:3]], want[bad[:3]])
    assert dbg["fast_order"] == 1 and dbg["fast_redo_rays"] > 0.2 * m, dbg
    assert np.array_equal(loop.trace_any(sc, rays), osc.trace_any(rays))
    # a frame through the integrator (bounce rays start ON these surfaces), and the same with the reference order forced
    cam = pkg.Camera.look_at((0.3, 1.5, 2.5), (0.0, 0.0, 0.0), 50.0, 16 / 9)
    img = osc.render_pass(cam, 128, 72, 0, 6)
    assert np.array_equal(bits(loop.render_pass(sc, cam, pkg.PathTracer(6), 128, 72, 0)), bits(img))
    monkeypatch.setenv("PTX_EXACT_ORDER", "1")
    lp = pkg.HipRenderLoop(0)
    assert np.array_equal(lp.trace_closest(sc, rays).view(np.uint8), want.view(np.uint8))
    assert lp.debug_stats()["fast_order"] == 0
    assert np.array_equal(bits(lp.render_pass(sc, cam, pkg.PathTracer(6), 128, 72, 0)), bits(img))
