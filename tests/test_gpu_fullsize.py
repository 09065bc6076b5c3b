"""GPU parity on the BASELINE.json workloads at their REAL size (detail 1.0), through the C ABI, against the det oracle.

The small-scene tests (test_gpu_parity.py) never reach the deep trees of the real scenes: there the 12-entry LDS stack
spills into the per-lane HBM strips, packets outgrow their shared stack and hand their rays to `k_extend_redo`, the
10 000-instance TLAS is 9+ levels deep, and a 4K frame is cut into several wavefront batches.  Every test below says,
through `ptx_get_debug_stats`, which of those roads its launches actually took.

  C2  sponza_like      ~300 k triangles, 117 instances, HDR sky + sun
  C3  bistro_like      ~2.6 M triangles, 1 181 instances, HDR sky + sun
  C4  instanced_stress 10 000 instances of 4 BLASes, 3840x2160
  C5  bistro_like + 10 % subsurface materials, uniform + adaptive passes

Bar: ids, t/u/v bits, any-hit flags, traversal counters and images bit-exact (integer / f32-bit equality).  The oracle
is parity-unpinned against the reference itself (no Rust toolchain, no fixtures: DESIGN.md §2).
"""
import numpy as np
import pytest

from oracle_py import OracleFramebuffer
from test_gpu_parity import bits, secondary_rays

pytestmark = pytest.mark.gpu

FULL = {
    # name: (flags, bounces)
    "sponza_like": (1, 16),
    "bistro_like": (3, 16),          # flags 3 = app lighting + 10 % SSS materials: C3's geometry, C5's materials
    "instanced_stress": (0, 4),
}
TRAVERSAL_COUNTERS = ("rays_closest", "rays_any", "node_visits", "node_children", "prim_index_reads", "triangle_tests",
                      "instance_entries", "instance_hits", "any_node_visits", "any_node_children", "any_triangle_tests",
                      "any_instance_entries")


@pytest.fixture(scope="module")
def full_scenes(pkg):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pkg.Scene.procedural(name, 1, 1.0, FULL[name][0])
        return cache[name]
    yield get
    cache.clear()


@pytest.mark.parametrize("name", list(FULL))
def test_full_detail_hit_records_bit_exact(pkg, loop, oracle_det, full_scenes, monkeypatch, name):
    """>= 100 k primary + bounce + finite-t_max rays through ptx_trace_batch, per-ray and as packets."""
    abi = pkg._abi
    sc = full_scenes(name)
    w, h = 480, 270
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    prim_c = osc.primary_rays(cam, w, h, 2)
    prim_g = loop.primary_rays(sc, cam, w, h, 2)
    assert np.array_equal(prim_c.view(np.uint8), prim_g.view(np.uint8))
    hc = osc.trace_closest(prim_c)
    sec = secondary_rays(abi, osc, prim_c, hc, 11)
    assert len(prim_c) + len(sec) >= 100_000
    monkeypatch.setenv("PTX_PACKET_BATCH", "1")
    packet_loop = pkg.HipRenderLoop(0)          # ptx_trace_batch through trace_packets
    monkeypatch.delenv("PTX_PACKET_BATCH")
    for label, rays in (("primary", prim_c), ("secondary", sec)):
        oracle_det.reset_counters()
        c = osc.trace_closest(rays)
        ca = osc.trace_any(rays)
        oc = oracle_det.counters()
        loop.set_counting(True)
        loop.reset_counters()
        g = loop.trace_closest(sc, rays)
        ga = loop.trace_any(sc, rays)
        gc = loop.counters()
        dbg = loop.debug_stats()
        loop.set_counting(False)
        assert np.array_equal(c["inst_idx"], g["inst_idx"]) and np.array_equal(c["tri_idx"], g["tri_idx"]), (name, label)
        for f in ("t", "u", "v"):
            assert np.array_equal(bits(c[f]), bits(g[f])), (name, label, f)
        assert np.array_equal(ca, ga), (name, label, "hit_any")
        for k in TRAVERSAL_COUNTERS:
            assert oc[k] == gc[k], (name, label, k, oc[k], gc[k])
        # uncounted kernels (the ones frames use) and the packet walk: same bytes
        assert np.array_equal(loop.trace_closest(sc, rays).view(np.uint8), c.view(np.uint8)), (name, label, "uncounted")
        packet_loop.reset_counters()
        assert np.array_equal(packet_loop.trace_closest(sc, rays).view(np.uint8), c.view(np.uint8)), (name, label, "packets")
        assert np.array_equal(packet_loop.trace_any(sc, rays), ca), (name, label, "any-hit packets")
        pdbg = packet_loop.debug_stats()
        print(f"\n[{name} {label}] {len(rays)} rays: per-ray stack high-water {dbg['trace_stack_high_water']} "
              f"(LDS holds {dbg['lds_stack_entries']}), packet stack high-water {pdbg['packet_stack_high_water']} "
              f"(LDS holds {pdbg['packet_lds_entries']}, capacity {pdbg['packet_capacity']}); "
              f"node visits/ray {gc['node_visits'] / max(1, gc['rays_closest'] + gc['rays_any']):.1f}")
        # the wave vote of the counted kernels: every action ran, and a lane-action count is the oracle's event count
        vc, va = dbg["vote_closest"], dbg["vote_any"]
        assert vc[0] > 0 and va[0] > 0 and all(vc[1 + k] <= vc[0] and 0 < vc[5 + k] <= 64 * vc[1 + k] for k in range(4)), (vc, va)
        assert vc[6] + va[6] == gc["triangle_tests"], (vc, va, gc["triangle_tests"])       # one TRI lane-action = one triangle test
        if name != "sponza_like" or label == "secondary":
            # the deep trees do leave the LDS part of the stack: the HBM strips are exercised
            assert dbg["trace_stack_high_water"] > dbg["lds_stack_entries"], (name, label, dbg)
    assert not oracle_det.stack_overflowed()


@pytest.mark.parametrize("name", list(FULL))
def test_full_detail_render_pass_bit_exact(pkg, loop, oracle_det, full_scenes, name):
    sc = full_scenes(name)
    flags, bounces = FULL[name]
    w, h = 256, 144
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    loop.reset_counters()
    for sample_index in (0, 7):
        c = osc.render_pass(cam, w, h, sample_index, bounces)
        g = loop.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, sample_index)
        assert np.isfinite(g).all()
        assert np.array_equal(bits(c), bits(g)), f"{name}: {np.mean((bits(c) != bits(g)).any(axis=2)):.4%} of pixels differ"
    dbg = loop.debug_stats()
    print(f"\n[{name} render_pass] packet stack high-water {dbg['packet_stack_high_water']}, redo rays {dbg['packet_redo_rays']}")
    # integrator + traversal event counts of one pass equal the oracle's
    oracle_det.reset_counters()
    osc.render_pass(cam, w, h, 1, bounces)
    oc = oracle_det.counters()
    loop.set_counting(True)
    loop.reset_counters()
    loop.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, 1)
    gc = loop.counters()
    dbg = loop.debug_stats()
    loop.set_counting(False)
    for k in TRAVERSAL_COUNTERS + ("material_fetches", "texture_lookups", "cdf_probes", "sky_pdf_evals", "sky_texel_lookups",
                                   "primary_rays", "pixel_samples_written", "sss_steps"):
        assert oc[k] == gc[k], (name, k, oc[k], gc[k])
    assert dbg["trace_stack_high_water"] > dbg["lds_stack_entries"], dbg
    if name == "bistro_like":
        assert gc["sss_steps"] > 0, "the SSS materials of config 5 were never hit"


@pytest.mark.parametrize("name", ["bistro_like", "instanced_stress"])
def test_full_detail_packet_overflow_takes_the_redo_road(pkg, oracle_det, full_scenes, monkeypatch, name):
    """With the packet stack cut to 12 entries the deep scenes push most primary packets into k_extend_redo; the frame
    must not change (and the test proves the redo kernel ran on a full-size tree)."""
    monkeypatch.setenv("PTX_PACKET_CAPACITY", "12")
    lp = pkg.HipRenderLoop(0)
    sc = full_scenes(name)
    flags, bounces = FULL[name]
    w, h = 256, 144
    cam = sc.default_camera(w / h)
    lp.reset_counters()
    g = lp.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, 3)
    dbg = lp.debug_stats()
    assert dbg["packet_redo_rays"] > 0.2 * w * h, dbg
    c = oracle_det.scene(sc).render_pass(cam, w, h, 3, bounces)
    assert np.array_equal(bits(c), bits(g))


@pytest.mark.parametrize("ray_width,packet_width", [(8, 8), (4, 0), (5, 3)])
def test_collapsed_trees_do_not_change_a_hit(pkg, oracle_det, full_scenes, monkeypatch, ray_width, packet_width):
    """The kernels may walk an order-preserving collapse of the reference's tree (csrc/device/pt_types.h: an internal
    child replaced in place by its own child list keeps the depth-first order of the leaves, which is all a result
    depends on).  By default only the packet kernels do (8 children per record); here the per-ray kernels walk
    collapsed trees too, at several widths, on the deep scenes — per-ray, as packets, and through a frame."""
    abi = pkg._abi
    monkeypatch.setenv("PTX_WIDE_RAY_BVH", str(ray_width))
    monkeypatch.setenv("PTX_WIDE_BVH", str(packet_width))
    lp = pkg.HipRenderLoop(0)
    monkeypatch.setenv("PTX_PACKET_BATCH", "1")
    lpk = pkg.HipRenderLoop(0)
    for name in ("bistro_like", "instanced_stress"):
        sc = full_scenes(name)
        w, h = 320, 180
        cam = sc.default_camera(w / h)
        osc = oracle_det.scene(sc)
        prim = osc.primary_rays(cam, w, h, 5)
        hc = osc.trace_closest(prim)
        for rays in (prim, secondary_rays(abi, osc, prim, hc, 3)):
            c, ca = osc.trace_closest(rays), osc.trace_any(rays)
            for loop_ in (lp, lpk):
                assert np.array_equal(loop_.trace_closest(sc, rays).view(np.uint8), c.view(np.uint8)), (name, ray_width, packet_width)
                assert np.array_equal(loop_.trace_any(sc, rays), ca), (name, ray_width, packet_width)
        img = osc.render_pass(cam, 128, 72, 2, FULL[name][1])
        assert np.array_equal(bits(lp.render_pass(sc, cam, pkg.PathTracer(FULL[name][1]), 128, 72, 2)), bits(img))
    # rays that can produce NaN slabs (zero direction components, origins on slab planes) take the faithful tree
    sc = pkg.Scene.procedural("cornell", 1, 1.0, 0)
    osc = oracle_det.scene(sc)
    grid = [(ox, oy, oz, *d) for ox in (-1.0, 0.0, 0.3, 1.0) for oy in (0.0, 0.6, 2.0) for oz in (-1.0, 0.0, 0.35, 2.5)
            for d in ((1, 0, 0), (0, -1, 0), (0, 0, 1), (1, 1, 0), (0, -1, 1), (-0.0, 0.0, -1.0), (0.3, -0.2, -1.0))]
    r = np.array(grid, dtype=np.float32)
    a = np.zeros(len(r), dtype=abi.ray_dtype())
    a["origin"], a["direction"], a["t_max"] = r[:, :3], r[:, 3:], np.inf
    for loop_ in (lp, lpk):
        assert np.array_equal(loop_.trace_closest(sc, a).view(np.uint8), osc.trace_closest(a).view(np.uint8))
        assert np.array_equal(loop_.trace_any(sc, a), osc.trace_any(a))


def test_c5_uniform_plus_adaptive_frame_bit_exact(pkg, loop, oracle_det, full_scenes):
    """BASELINE config 5 on its real scene: uniform passes (a pixel-major batch), then adaptive passes in flight,
    against the oracle's pass-at-a-time AdaptiveFramebuffer."""
    sc = full_scenes("bistro_like")
    w, h, spp, cap, bounces = 192, 108, 6, 14, 16
    cam = sc.default_camera(w / h)
    cfg = pkg.AdaptiveSamplingConfig(min_samples=4, variance_threshold=0.05)
    mean, counts, stats = loop.render_frame(sc, cam, pkg.PathTracer(bounces), w, h, spp, adaptive=cfg, max_spp=cap)
    ofb = OracleFramebuffer(oracle_det.scene(sc), w, h, True)
    ofb.render_passes(cam, 0, spp, bounces)
    passes = samples = 0
    while ofb.read()[2].max() < cap:
        n = ofb.adaptive_pass(cam, 4, 0.05, w * h, bounces)
        if n == 0:
            break
        passes += 1
        samples += n
    oc = ofb.read()[2]
    assert passes >= 3 and oc.min() < oc.max()
    assert stats["adaptive_passes"] == passes and stats["adaptive_samples"] == samples
    assert np.array_equal(counts, oc) and np.array_equal(bits(mean), bits(ofb.mean()))


def test_c4_4k_frame_in_several_wavefront_batches(pkg, loop, oracle_det, full_scenes, monkeypatch):
    """3840x2160 on the 10 000-instance TLAS with the path budget cut so that 7 passes run as 3 + 3 + 1 wavefront
    batches (the default 128 Mi slots hold 16 passes of a 4K frame: a 32 spp frame is two batches).  Size-independent
    properties: the batching does not change a bit, run-to-run determinism, counts == spp, clamped finite samples,
    partition additivity."""
    sc = full_scenes("instanced_stress")
    w, h, spp, bounces = 3840, 2160, 7, 4
    cam = sc.default_camera(w / h)
    r = pkg.PathTracer(bounces)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    fb.render_passes(sc, cam, r, 0, spp)          # one batch of 7
    s1, q1, c1 = fb.read()
    assert (c1 == spp).all() and np.isfinite(s1).all() and s1.min() >= 0.0 and s1.max() <= 10.0 * spp * (1 + 1e-6)
    assert fb.stats() == (spp, spp, spp * w * h)
    monkeypatch.setenv("PTX_PATHS_IN_FLIGHT", str(3 * w * h))
    lp = pkg.HipRenderLoop(0)
    fb3 = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
    fb3.render_passes(sc, cam, r, 0, spp)         # 3 + 3 + 1
    s3, q3, c3 = fb3.read()
    assert np.array_equal(c1, c3) and np.array_equal(bits(s1), bits(s3)) and np.array_equal(bits(q1), bits(q3)), \
        "the split into wavefront batches changed the sums"
    fb3.reset()
    fb3.render_passes(sc, cam, r, 0, spp)
    s3b = fb3.read()[0]
    assert np.array_equal(bits(s3), bits(s3b)), "not deterministic run to run"
    try:
        lp.set_partition(2, 3)
        fb3.reset()
        fb3.render_passes(sc, cam, r, 0, spp)
        sp, _, cp = fb3.read()
    finally:
        lp.set_partition(0, 1)
    own = cp > 0
    assert abs(own.mean() - 1 / 3) < 0.01 and np.array_equal(bits(sp[own]), bits(s1[own])) and (sp[~own] == 0).all()
    # the 4K primaries themselves: every 61st of the 8.3 M rays against the oracle
    osc = oracle_det.scene(sc)
    all_rays = lp.primary_rays(sc, cam, w, h, 0)
    assert np.array_equal(all_rays.view(np.uint8), osc.primary_rays(cam, w, h, 0).view(np.uint8))
    rays = np.ascontiguousarray(all_rays[::61])
    assert len(rays) > 100_000
    assert np.array_equal(lp.trace_closest(sc, rays).view(np.uint8), osc.trace_closest(rays).view(np.uint8))


def test_dof_camera_and_rotated_sky_bit_exact(pkg, loop, oracle_det):
    """The two corners no other test reaches: `generate_ray_dof` with aperture > 0 (camera.rs:81-94) and a sky with
    rotation != 0 (sky/texture.rs:41-65: rotate_direction / unrotate_direction around Y)."""
    sc = pkg.Scene.procedural("cornell", 1, 1.0, 1)        # HDR sky + sun
    sc.desc.contents.sky.rotation = 0.7                     # both sides read the same descriptor
    w, h = 96, 54
    cam = pkg.Camera.from_angles((0.0, 1.0, 3.0), 0.0, -0.15, 60.0, w / h, aperture=0.08, focus_distance=2.5)
    assert cam.aperture > 0
    osc = oracle_det.scene(sc)
    lp = pkg.HipRenderLoop(0)                               # uploads the mutated descriptor
    pc, pg = osc.primary_rays(cam, w, h, 4), lp.primary_rays(sc, cam, w, h, 4)
    assert np.array_equal(pc.view(np.uint8), pg.view(np.uint8)), "DoF primary rays differ"
    assert (osc.trace_closest(pc)["inst_idx"] != pkg._abi.PTX_MISS).mean() > 0.3, "the camera does not look at the box"
    pinhole = pkg.Camera.from_angles((0.0, 1.0, 3.0), 0.0, -0.15, 60.0, w / h)
    assert not np.array_equal(osc.primary_rays(pinhole, w, h, 4)["origin"], pc["origin"]), "the DoF branch did not run"
    for s in (0, 4):
        assert np.array_equal(bits(osc.render_pass(cam, w, h, s, 8)), bits(lp.render_pass(sc, cam, pkg.PathTracer(8), w, h, s)))
    flat = pkg.Scene.procedural("cornell", 1, 1.0, 1)
    assert not np.array_equal(oracle_det.scene(flat).render_pass(cam, w, h, 0, 8), osc.render_pass(cam, w, h, 0, 8)), "rotation had no effect"


def test_zero_and_oversized_bounce_limits(pkg, loop, oracle_det):
    """`PathTracer<0>`: the bounce loop never runs, every sample is black and still counts; > 255 is refused."""
    sc = pkg.Scene.procedural("cornell", 1, 1.0, 0)
    w, h = 32, 18
    cam = sc.default_camera(w / h)
    g = loop.render_pass(sc, cam, pkg.PathTracer(0), w, h, 0)
    assert np.array_equal(bits(g), bits(oracle_det.scene(sc).render_pass(cam, w, h, 0, 0))) and (g == 0).all()
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    fb.render_passes(sc, cam, pkg.PathTracer(0), 0, 3)
    s, _, c = fb.read()
    assert (c == 3).all() and (s == 0).all()
    with pytest.raises(pkg.RenderCoreError):
        loop.render_pass(sc, cam, pkg.PathTracer(256), w, h, 0)


def test_forwarded_bound_changes_no_hit_and_hits_sit_inside_their_boxes(pkg, loop, oracle_det, full_scenes, monkeypatch):
    """Closest-hit rays take the world closest_t (x 1.03125 + the certified rounding bound of world -> local -> world)
    into an instance instead of the reference's +inf (`forwarded_bound`, csrc/device/pt_traverse.h; DESIGN.md says why
    the comparison after the instance rejects everything the tighter start skips).  The argument presumes that a
    triangle hit does not lie far in front of the boxes above it; the counted kernels measure exactly that
    (`max_leaf_undershoot`: the bound tolerates 2^-5, here it must stay below 2^-9), and with forwarding switched off
    (PTX_NO_FORWARD=1, read at upload) the same rays must give the same bytes."""
    abi = pkg._abi
    monkeypatch.setenv("PTX_NO_FORWARD", "1")
    lp_off = pkg.HipRenderLoop(0)
    for name in FULL:
        sc = full_scenes(name)
        w, h = 400, 225
        cam = sc.default_camera(w / h)
        osc = oracle_det.scene(sc)
        prim = osc.primary_rays(cam, w, h, 9)
        rays = np.concatenate([prim, secondary_rays(abi, osc, prim, osc.trace_closest(prim), 21)])
        want = osc.trace_closest(rays)
        off = lp_off.trace_closest(sc, rays)                 # uploads with the switch set
        assert lp_off.debug_stats()["bound_forwarding"] == 0
        monkeypatch.delenv("PTX_NO_FORWARD")
        loop._scene = None                                   # a fresh upload without the switch
        on = loop.trace_closest(sc, rays)
        assert loop.debug_stats()["bound_forwarding"] == 1, name
        monkeypatch.setenv("PTX_NO_FORWARD", "1")
        assert np.array_equal(on.view(np.uint8), want.view(np.uint8)) and np.array_equal(off.view(np.uint8), want.view(np.uint8)), name
        loop.set_counting(True)
        loop.reset_counters()
        loop.trace_closest(sc, rays)
        loop.trace_any(sc, rays)
        dbg = loop.debug_stats()
        loop.set_counting(False)
        print(f"\n[{name}] {len(rays)} rays: largest (t_enter(leaf) - t) / t over all triangle hits = {dbg['max_leaf_undershoot']:.3e}")
        assert 0.0 <= dbg["max_leaf_undershoot"] < 2.0 ** -9, (name, dbg)
        assert dbg["max_leaf_undershoot_excess"] < 2.0 ** -13, (name, dbg)      # beyond the scene's absolute slack: what the nearest-first walk's 2^-10 must cover
        img = osc.render_pass(cam, 160, 90, 4, FULL[name][1])
        assert np.array_equal(bits(loop.render_pass(sc, cam, pkg.PathTracer(FULL[name][1]), 160, 90, 4)), bits(img)), name


def test_side_stream_overlap_changes_nothing(pkg, oracle_det, full_scenes, monkeypatch):
    """By default an iteration's any-hit launch and k_finalize run on a second stream beside the next iteration's
    closest-hit launch (run_paths).  The frame must be the one-stream frame bit for bit — plain passes, several batches
    in a row, adaptive passes and the subsurface walk queue — and equal the oracle's."""
    sc = full_scenes("bistro_like")           # flags 3: sky + sun + SSS materials
    w, h, spp, bounces = 192, 108, 6, 16
    cam = sc.default_camera(w / h)
    frames = []
    monkeypatch.setenv("PTX_TAIL_CAP", "3000")      # k_tail (which has no side stream) only takes the last 3000 paths of a batch
    for overlap in ("1", "0"):
        monkeypatch.setenv("PTX_OVERLAP_SHADOW", overlap)
        monkeypatch.setenv("PTX_PATHS_IN_FLIGHT", str(w * h * 4))       # 6 passes = a batch of 4 + a batch of 2
        lp = pkg.HipRenderLoop(0)
        fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
        fb.render_passes(sc, cam, pkg.PathTracer(bounces), 0, spp)
        lp.render_adaptive_passes(sc, cam, pkg.PathTracer(bounces), fb, pkg.AdaptiveSamplingConfig(4, 0.0005), 3)
        frames.append(fb.read())
        one = lp.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, 2)
        frames[-1] = frames[-1] + (one,)
    for a, b in zip(frames[0], frames[1]):
        assert np.array_equal(bits(a), bits(b))
    assert np.array_equal(bits(frames[0][3]), bits(oracle_det.scene(sc).render_pass(cam, w, h, 2, bounces)))


def test_sky_search_ahead_changes_nothing(pkg, oracle_det, full_scenes, monkeypatch):
    """By default k_shade starts the sky-light sample's table searches ahead of the hit-data gathers, on the numbers the
    sample will draw unless the material draws first (`k_shade<false, true>`, `gather_hit_and_sky`); vertices on glass or
    subsurface materials, which do draw first, search again with their own numbers.  The frame must be the
    `PTX_SKY_AHEAD=0` frame bit for bit — plain passes, adaptive passes, the subsurface walk queue — and equal the
    oracle's.  `PTX_TAIL_CAP` keeps the paths in k_shade (k_tail's shading phase never searches ahead)."""
    sc = full_scenes("bistro_like")           # flags 3: HDR sky + sun + glass and SSS materials
    w, h, spp, bounces = 192, 108, 5, 16
    cam = sc.default_camera(w / h)
    frames = []
    monkeypatch.setenv("PTX_TAIL_CAP", "3000")
    for ahead in ("1", "0"):
        monkeypatch.setenv("PTX_SKY_AHEAD", ahead)
        lp = pkg.HipRenderLoop(0)
        fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
        fb.render_passes(sc, cam, pkg.PathTracer(bounces), 0, spp)
        lp.render_adaptive_passes(sc, cam, pkg.PathTracer(bounces), fb, pkg.AdaptiveSamplingConfig(4, 0.0005), 2)
        frames.append(fb.read() + (lp.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, 3),))
    for a, b in zip(frames[0], frames[1]):
        assert np.array_equal(bits(a), bits(b))
    assert np.array_equal(bits(frames[0][3]), bits(oracle_det.scene(sc).render_pass(cam, w, h, 3, bounces)))


def test_tail_kernel_changes_nothing(pkg, oracle_det, full_scenes, monkeypatch):
    """By default the late iterations of a batch run as one k_tail launch (every wave carries its own paths through
    extend -> shade -> walk step -> shadow -> finalize until they end; run_paths).  The frame must be the
    launch-per-iteration frame bit for bit: small batches (k_tail from the first shading on, sized from the batch alone), large
    ones (the takeover point found by reading the live count back), few and many paths per wave, plain passes, adaptive
    passes, the subsurface walk queue — and equal the oracle's."""
    sc = full_scenes("bistro_like")           # flags 3: sky + sun + SSS materials
    bounces = 16
    results = {}
    for label, env, (w, h, spp) in (("off", {"PTX_TAIL_KERNEL": "0"}, (192, 108, 6)),
                                   ("on", {}, (192, 108, 6)),
                                   ("on/8 per wave", {"PTX_TAIL_MIN_PER_WAVE": "8"}, (192, 108, 6)),
                                   ("on/254 per wave", {"PTX_TAIL_MIN_PER_WAVE": "254"}, (192, 108, 6)),
                                   ("on/late", {"PTX_TAIL_CAP": "3000"}, (192, 108, 6)),
                                   ("on/late, one stream", {"PTX_TAIL_CAP": "3000", "PTX_OVERLAP_SHADOW": "0"}, (192, 108, 6)),
                                   ("off/large", {"PTX_TAIL_KERNEL": "0"}, (640, 360, 4)),
                                   ("on/large", {"PTX_TAIL_FIRST_POLL": "2"}, (640, 360, 4))):
        for k in ("PTX_TAIL_KERNEL", "PTX_TAIL_MIN_PER_WAVE", "PTX_TAIL_FIRST_POLL", "PTX_TAIL_CAP", "PTX_OVERLAP_SHADOW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("PTX_PATHS_IN_FLIGHT", str(w * h * 4))       # 6 passes = a batch of 4 + a batch of 2
        lp = pkg.HipRenderLoop(0)
        lp.set_profiling(True)
        cam = sc.default_camera(w / h)
        fb = pkg.DeviceFramebuffer(lp, w, h, adaptive=True)
        fb.render_passes(sc, cam, pkg.PathTracer(bounces), 0, spp)
        lp.render_adaptive_passes(sc, cam, pkg.PathTracer(bounces), fb, pkg.AdaptiveSamplingConfig(4, 0.0005), 3)
        one = lp.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, 2)
        t = lp.timings()
        assert (t["tail_launches"] > 0) == label.startswith("on"), (label, t)
        print(f"\n[{label}] {w}x{h}: tail launches {t['tail_launches']}, {t['tail_ms']:.2f} ms; extend launches {t['extend_launches']}, {t['extend_ms']:.2f} ms")
        results[label] = fb.read() + (one,)
    for label in ("on", "on/8 per wave", "on/254 per wave", "on/late", "on/late, one stream"):
        for a, b in zip(results["off"], results[label]):
            assert np.array_equal(bits(a), bits(b)), label
    for a, b in zip(results["off/large"], results["on/large"]):
        assert np.array_equal(bits(a), bits(b))
    assert np.array_equal(bits(results["on"][3]), bits(oracle_det.scene(sc).render_pass(sc.default_camera(192 / 108), 192, 108, 2, bounces)))


def test_nearest_first_equals_reference_order_on_the_full_scenes(pkg, loop, oracle_det, full_scenes, monkeypatch):
    """FAST mode (nearest child first + the second attempt in reference order for rays with a near-tied second hit) on
    by default: hit records and frames must equal both the oracle's and the same kernels with PTX_EXACT_ORDER=1; the
    share of rays that needed the second attempt is printed (1.4 % of the Bistro-like scene's bounce rays)."""
    abi = pkg._abi
    monkeypatch.setenv("PTX_EXACT_ORDER", "1")
    lp_ref = pkg.HipRenderLoop(0)
    for name in FULL:
        sc = full_scenes(name)
        w, h = 400, 225
        cam = sc.default_camera(w / h)
        osc = oracle_det.scene(sc)
        prim = osc.primary_rays(cam, w, h, 13)
        sec = secondary_rays(abi, osc, prim, osc.trace_closest(prim), 31)
        want = osc.trace_closest(sec)
        ref = lp_ref.trace_closest(sc, sec)                  # uploads with the switch set
        assert lp_ref.debug_stats()["fast_order"] == 0
        monkeypatch.delenv("PTX_EXACT_ORDER")
        loop._scene = None
        loop.reset_counters()
        got = loop.trace_closest(sc, sec)
        dbg = loop.debug_stats()
        monkeypatch.setenv("PTX_EXACT_ORDER", "1")
        assert dbg["fast_order"] == 1, name
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)) and np.array_equal(ref.view(np.uint8), want.view(np.uint8)), name
        print(f"\n[{name}] {len(sec)} bounce rays and segments: {dbg['fast_redo_rays']} second attempts ({dbg['fast_redo_rays'] / len(sec):.2%})")
        assert dbg["fast_redo_rays"] < 0.2 * len(sec), dbg
        img = osc.render_pass(cam, 160, 90, 6, FULL[name][1])
        assert np.array_equal(bits(loop.render_pass(sc, cam, pkg.PathTracer(FULL[name][1]), 160, 90, 6)), bits(img)), name
        assert np.array_equal(bits(lp_ref.render_pass(sc, cam, pkg.PathTracer(FULL[name][1]), 160, 90, 6)), bits(img)), name
