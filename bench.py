#!/usr/bin/env python3
"""bench.py — the reference's headline measurement on MI355X.

Metric (BASELINE.json): Mrays/s + s/frame, Bistro 1080p 64 spp, 1/2/4/8 MI355X,
% of the HBM roofline, next to the host-CPU path.

A "step" is one whole frame: `for sample in 0..spp { render_pass; accumulate }`
(main.rs:124-133) on the device-resident framebuffer, scene already in HBM.
With N > 1 the frame is split by interleaved 8x8 pixel tiles (scene replicated)
and the per-pixel sum / sum_sq / count records of every rank's own tiles are
exchanged with ONE RCCL all-gather per frame and scattered into each rank's
framebuffer (ownership is disjoint, so the combined buffers are bit-identical
to the single-GPU ones).  Total work is
fixed as N grows: "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def extend_bytes(c):
    """Algorithmic bytes of the closest-hit (extend) traversal only: SURVEY.md §8(d) terms
    node + prim-index + triangle + instance, with the any-hit share removed."""
    nv = c["node_visits"] - c["any_node_visits"]
    nc = c["node_children"] - c["any_node_children"]
    pr = c["prim_index_reads"] - c["any_prim_index_reads"]
    tt = c["triangle_tests"] - c["any_triangle_tests"]
    ie = c["instance_entries"] - c["any_instance_entries"]
    return 4 * nv + 28 * nc + 4 * pr + 48 * tt + 64 * ie + 128 * c["instance_hits"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="bistro_like")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=16)      # PathTracer<16> of the app (app.rs:152)
    ap.add_argument("--flags", type=int, default=1)         # app lighting: HDR sky IS + sun
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    # frames kept in flight during the timed region: 2 = two contexts render alternate frames (ptx_fb_render_passes_async +
    # ptx_order_after_frame_tail), so the latency-bound last iterations of one frame share the GPU with the first iterations
    # of the next; 1 = one frame at a time (default).  Measured: no gain on a whole frame per GPU (564.3 vs
    # 564.5 ms), 83.5 -> 80.4 ms on an eighth of the frame (predicted 8-GPU efficiency 0.84 -> 0.88); the per-kernel event
    # times of overlapping frames are no longer each kernel's own, so the roofline line is quoted with 1.
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=(1, 2))
    ap.add_argument("--no-order-check", action="store_true")   # skip the untimed reference-order frame (kernel experiments only)
    # one-GPU prediction of an N-GPU frame: render every rank's tile partition in turn on this card (no collective)
    ap.add_argument("--emulate-world", type=int, default=0)
    ap.add_argument("--adaptive-cap", type=int, default=0)  # with --emulate-world: adaptive phase up to this many spp (BASELINE config 5: 256)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render core has no CPU fallback")
    # Rehearsal on a one-GPU box only (never the driver's launch): PTX_BENCH_DEVICE pins every rank to one card and
    # PTX_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one GPU; the control flow is the same.
    if os.environ.get("PTX_BENCH_DEVICE"):
        local_rank = int(os.environ["PTX_BENCH_DEVICE"])
    backend = os.environ.get("PTX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # RCCL
        else:
            dist.init_process_group(backend=backend)

    verbose = bool(os.environ.get("PTX_BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr)
    note("process group up" if world > 1 else "single rank")
    pkg = ge.load_package()
    if os.environ.get("PT_VARIANT_LIB"):      # kernel experiments only: a library built by tools/build_variant.sh
        pkg.render.DEVICE_LIB_PATH = os.path.abspath(os.environ["PT_VARIANT_LIB"])
    w, h, spp = args.width, args.height, args.spp
    t0 = time.time()
    scene = pkg.Scene.procedural(args.scene, args.seed, args.detail, args.flags)
    build_s = time.time() - t0
    cam = scene.default_camera(w / h)
    renderer = pkg.PathTracer(args.bounces)
    loop = pkg.HipRenderLoop(local_rank)
    loop.set_partition(rank, world)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    loop._bind(scene, cam)   # upload once: inputs are resident in HBM before anything is timed
    note(f"scene built in {build_s:.1f} s and uploaded")

    # contexts that render alternate frames of the timed region (each with its own path state, queues, streams and framebuffer)
    nfl = args.frames_in_flight if args.steps > 1 else 1
    loops, fbs = [loop], [fb]
    for _ in range(1, nfl):
        lp_k = pkg.HipRenderLoop(local_rank)
        lp_k.set_partition(rank, world)
        fbs.append(pkg.DeviceFramebuffer(lp_k, w, h, adaptive=True))
        lp_k._bind(scene, cam)
        loops.append(lp_k)
    exchanges = [pkg.render.TileExchange(dist, lp_k, w, h, rank, world, host_staged=(backend != "nccl")) for lp_k in loops] if world > 1 else None

    def frame(r=None, k=0, wait=True):
        if nfl > 1:
            # this context's frame starts when the other context's frame is past its heavy iterations: the two frames'
            # big launches do not compete, the other frame's latency-bound tail runs beside this frame's start
            loops[k].order_after_frame_tail(loops[1 - k])
        fbs[k].reset()
        fbs[k].render_passes(scene, cam, r or renderer, 0, spp, wait=wait)
        if world > 1:
            # the one exchange step of the frame: all-gather of every rank's owned tiles (RCCL over xGMI), scattered
            # into each rank's framebuffer, which then holds the whole frame (bit-identical to the 1-GPU frame)
            exchanges[k].combine(fbs[k])

    def fence():
        for lp_k in loops:
            lp_k.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # counted frames (untimed; counted kernels are slower): rays and algorithmic bytes per frame.
    #   mode 1: the counted kernels walk exactly as the reference does -> the REFERENCE's events (equal the oracle's);
    #   mode 2: the closest-hit launches walk as the timed kernels do (forwarded bound, nearest child first, second
    #           attempts included) -> the events the TIMED per-ray kernel executes: the roofline numerator.
    # Each also cut after its primary rays (PathTracer<1>): the closest-hit events of the first extend launch, which the
    # timed frames trace with the packet kernel; the per-ray kernel's bytes are the rest.
    def counted(mode, r=None):
        loop.set_counting(mode)
        loop.reset_counters()
        frame(r)
        fence()
        c = loop.counters()
        if world > 1:
            keys = sorted(c)
            t = torch.tensor([c[k] for k in keys], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            c = {k: int(v) for k, v in zip(keys, t.tolist())}
        return c

    cnt = counted(1)
    dbg_counted = loop.debug_stats()      # max_leaf_undershoot over every triangle hit of the counted frame: the margins' presumption, measured
    cnt_primary = counted(1, pkg.PathTracer(1))
    cnt_ship = counted(2)
    cnt_ship_primary = counted(2, pkg.PathTracer(1))
    loop.set_counting(0)
    note("counted frames done")

    for i in range(args.warmup):
        frame(k=i % nfl)
        note("warm-up frame done")
    if args.warmup:
        for k in range(args.warmup, nfl):       # every context has rendered once before the clock starts (its allocations are made)
            frame(k=k)
    fence()
    note("fenced")
    for lp_k in loops:
        lp_k.reset_counters()
        lp_k.set_profiling(True)
        lp_k.reset_timings()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(k=i % nfl, wait=(nfl == 1))
    fence()
    elapsed = time.perf_counter() - t0
    note(f"timed region {elapsed:.3f} s")
    tim = {}
    for lp_k in loops:                          # HIP events of every context's launches
        for key, v in lp_k.timings().items():
            tim[key] = tim.get(key, 0) + v
        lp_k.set_profiling(False)
    fb = fbs[(args.steps - 1) % nfl]            # the last timed frame is the one that checks itself below
    loop_last = loops[(args.steps - 1) % nfl]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the timed frames check themselves (outside the timed region) ----------------------------------------
    # fb.stats() goes through ptx_fb_stats, which reports a traversal stack overflow in any launch since the last
    # check (a ray cut short is less work: a frame like that must not produce a number)
    mn, mx, total = fb.stats()
    s_sum, s_sq, s_cnt = fb.read()
    owned = s_cnt > 0
    n_owned = int(owned.sum())
    want_owned = w * h          # after the exchange every rank's framebuffer holds the whole frame
    verify = {
        "status": "PTX_OK",
        "counts_eq_spp": bool(mn == spp and mx == spp and n_owned == want_owned and (s_cnt[owned] == spp).all()),
        "finite": bool(np.isfinite(s_sum).all() and np.isfinite(s_sq).all() and float(s_sum.min()) >= 0.0),
        "clamped": bool(float(s_sum.max()) <= 10.0 * spp * (1 + 1e-6)),          # clamp_radiance: max component <= 10
        "mean_radiance": round(float(s_sum.sum(dtype=np.float64)) / max(1, n_owned * spp * 3), 6),
        "sum_sha256_16": __import__("hashlib").sha256(s_sum.tobytes()).hexdigest()[:16],
    }
    # what the forwarded bound and the nearest-first walk presume, measured on this run: how far a computed
    # triangle hit lay in front of its leaf box at most (counted frame, relative), and how many closest-hit rays of the
    # timed frames were traced a second time in reference order
    dbg_timed = loop_last.debug_stats()
    dbg_timed["fast_redo_rays"] = sum(lp_k.debug_stats()["fast_redo_rays"] for lp_k in loops)
    verify["max_leaf_undershoot"] = float(f"{dbg_counted['max_leaf_undershoot']:.3e}")
    verify["max_leaf_undershoot_beyond_slack"] = float(f"{dbg_counted['max_leaf_undershoot_excess']:.3e}")
    verify["undershoot_tolerated"] = {"forwarded_bound": 2.0 ** -5, "nearest_first": 2.0 ** -10}
    verify["bound_forwarding"] = int(dbg_timed["bound_forwarding"])
    verify["nearest_first"] = int(dbg_timed["fast_order"])
    verify["second_attempt_rays_per_frame"] = int(dbg_timed["fast_redo_rays"] // max(1, args.steps))
    verify["uncertified_meshes"] = int(dbg_timed["uncertified_meshes"])
    if not (verify["counts_eq_spp"] and verify["finite"] and verify["clamped"]):
        raise SystemExit(f"bench.py: the timed frame failed its own checks: {verify}")
    # The timed kernels take two shortcuts the reference's walk does not (forwarded bound, nearest child first).  One
    # more UNTIMED frame in a second context that takes neither (PTX_EXACT_ORDER=1 PTX_NO_FORWARD=1: the reference's
    # walk operation for operation) must give the same sums bit for bit, or no number is printed.
    if not args.no_order_check:
        saved = {k: os.environ.get(k) for k in ("PTX_EXACT_ORDER", "PTX_NO_FORWARD")}
        os.environ["PTX_EXACT_ORDER"] = "1"
        os.environ["PTX_NO_FORWARD"] = "1"
        try:
            loop2 = pkg.HipRenderLoop(local_rank)
            loop2.set_partition(rank, world)
            fb2 = pkg.DeviceFramebuffer(loop2, w, h, adaptive=True)
            fb2.reset()
            fb2.render_passes(scene, cam, renderer, 0, spp)
            loop2.sync()
            r_sum, r_sq, r_cnt = fb2.read()
            dbg2 = loop2.debug_stats()
            loop2.close()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        mine = r_cnt > 0                  # this rank's own tiles (the timed framebuffer holds every rank's after the exchange)
        same = bool(dbg2["fast_order"] == 0 and dbg2["bound_forwarding"] == 0 and int(mine.sum()) > 0 and
                    np.array_equal(s_sum[mine].view(np.uint32), r_sum[mine].view(np.uint32)) and
                    np.array_equal(s_sq[mine].view(np.uint32), r_sq[mine].view(np.uint32)) and np.array_equal(s_cnt[mine], r_cnt[mine]))
        if world > 1:
            t = torch.tensor([1 if same else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            same = bool(t.item())
        verify["equals_reference_order"] = same
        verify["reference_order_sum_sha256_16"] = __import__("hashlib").sha256(r_sum.tobytes()).hexdigest()[:16] if world == 1 else None
        if not same:
            raise SystemExit(f"bench.py: the timed frame differs from the reference-order frame (PTX_EXACT_ORDER=1 PTX_NO_FORWARD=1): {verify}")
        note("reference-order frame equals the timed frame")

    sec_per_frame = elapsed / args.steps
    rays_per_frame = cnt["rays_closest"] + cnt["rays_any"]
    primary_per_frame = w * h * spp
    mrays = rays_per_frame / sec_per_frame / 1e6
    # algorithmic bytes (SURVEY 8d formula) of the frame: with the timed walk's own traversal events, and with the reference's
    abytes_frame = pkg._abi.algorithmic_bytes(cnt_ship, texel_bytes_per_lookup=48, adaptive=True)
    abytes_frame_ref = pkg._abi.algorithmic_bytes(cnt, texel_bytes_per_lookup=48, adaptive=True)

    # roofline of the dominant kernel (k_extend), rank 0's launches, HIP events on the library stream
    ext_launches = max(1, tim["extend_launches"])
    ext_ms = tim["extend_ms"] / ext_launches
    # k_extend's launches are every iteration but those traced as packets (the primary rays, k_extend_packet):
    # its bytes are the frame's closest-hit bytes minus the primary launch's
    pk_launches = int(tim["packet_launches"])
    per_ray_bytes = extend_bytes(cnt_ship) - (extend_bytes(cnt_ship_primary) if pk_launches else 0)
    per_ray_bytes_ref = extend_bytes(cnt) - (extend_bytes(cnt_primary) if pk_launches else 0)
    # rank 0 traced 1/world of the frame's extend work (tile interleave): per-launch bytes on this rank
    ext_bytes_launch = per_ray_bytes / world * args.steps / ext_launches
    ext_bytes_launch_ref = per_ray_bytes_ref / world * args.steps / ext_launches
    # The counted events include the closest-hit passes of the paths k_tail carries (the last ~1 % of a frame's rays,
    # which the counted run traces with k_extend): their time is inside k_tail's, so ALL of k_tail's time (its shading,
    # any-hit and accumulation phases too) is charged to the stream.  That understates `achieved` a little, never overstates it.
    tail_ms_launch = tim.get("tail_ms", 0.0) / ext_launches
    achieved = ext_bytes_launch / ((ext_ms + tail_ms_launch) * 1e-3) / 1e9 if ext_ms > 0 else 0.0
    achieved_ref = ext_bytes_launch_ref / ((ext_ms + tail_ms_launch) * 1e-3) / 1e9 if ext_ms > 0 else 0.0
    # `achieved` / `frac`: the SURVEY 8(d) byte formula applied to the events the TIMED kernel executes (its own node visits,
    # child boxes, triangle tests, instance entries: ptx_set_counting(2)), over its measured launch time.  Every ray is
    # charged for every record it touches; L1/L2 serve most of that stream, so this is the kernel's record-stream rate
    # against the HBM peak, not HBM utilisation: `traffic` / `measured_hbm_frac` (PMC counters) are that, and
    # `binding_limiter` names what actually limits the kernel.
    roofline = {"bound": "hbm", "kernel": "k_extend<false, true>" if verify["nearest_first"] else "k_extend<false, false>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "measured_hbm_frac": None,
                "frac_meaning": "SURVEY 8d bytes of the events the timed kernel itself executes (counted instantiation of the same walk: forwarded bound, nearest child first, second attempts) / launch time / 8 TB/s; caches serve most of this stream, see measured_hbm_frac (PMC) for HBM utilisation and binding_limiter for what binds",
                "algorithmic_bytes_per_launch": round(ext_bytes_launch), "avg_launch_ms": round(ext_ms, 4),
                "launches": int(ext_launches),
                "tail_kernel": {"launches": int(tim.get("tail_launches", 0)), "ms_per_frame": round(tim.get("tail_ms", 0.0) / args.steps, 3),
                                "note": "k_tail carries the last paths of a batch through all their phases; the bytes above include its closest-hit passes, "
                                        "so achieved = bytes / (k_extend time + all of k_tail's time)"},
                # secondary: the same launches priced with the REFERENCE's walk (what round 2 reported as frac; can exceed 1
                # because the timed kernel skips part of that walk)
                "reference_stream": {"algorithmic_bytes_per_launch": round(ext_bytes_launch_ref), "achieved": round(achieved_ref, 2),
                                     "frac": round(achieved_ref / HBM_PEAK_GBS, 5)}}
    if pk_launches:
        pk_ms = tim["packet_ms"] / pk_launches
        pk_bytes = extend_bytes(cnt_primary) / world * args.steps / pk_launches      # reference stream: the packet kernel has no counted form
        roofline["primary_packets"] = {"kernel": "k_extend_packet", "launches": pk_launches, "avg_launch_ms": round(pk_ms, 4),
                                       "algorithmic_bytes_per_launch": round(pk_bytes),
                                       "achieved": round(pk_bytes / (pk_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                                       "note": "reference byte stream of the primary rays (one record fetch serves the 64 rays of a packet: not comparable with the HBM peak)"}
    # PMC-derived figures come from the committed profile of ONE configuration: attach them only when this run is that
    # configuration (otherwise a Cornell run would carry Bistro's traffic)
    this_config = {"scene": args.scene, "detail": args.detail, "width": w, "height": h, "spp": spp, "bounces": args.bounces,
                   "flags": args.flags, "seed": args.seed, "world": world}
    lim_file = os.path.join(ROOT, "profiles", "r03_limiters.json")
    if os.path.exists(lim_file):
        try:
            lim = json.load(open(lim_file))
            kernels = lim.get("kernels", {})
            want = "k_extend<false, true" if verify["nearest_first"] else "k_extend<false, false"      # nearest-first / reference-order instantiation
            k = next((v for n, v in kernels.items() if n.startswith(want)), None)
            if lim.get("config") == this_config and k:
                roofline["traffic"] = k["hbm_bytes_per_launch"]
                # counter bytes over THIS run's launch time
                roofline["measured_hbm_frac"] = round(k["hbm_bytes_per_launch"] / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                roofline["limiters"] = {n: k[n] for n in ("hbm_traffic_frac", "l2_bw_frac", "l2_hit_rate", "valu_issue_frac", "valu_issue_frac_at_peak_clock",
                                                         "clock_GHz_measured", "lanes_active", "mem_wait_share", "issue_wait_share", "salu_per_valu", "salu_issue_frac", "binding") if n in k}
                roofline["limiters"]["source"] = lim.get("source")
                # what actually binds the kernel (the contract's `bound` stays "hbm": achieved / peak are bytes per second)
                roofline["binding_limiter"] = {"name": k.get("binding"), "frac": k.get(k.get("binding", ""), None),
                                               "clock_GHz": k.get("clock_GHz_measured"),
                                               "note": "VALU issue: wave64 VALU instructions x 2 clocks / (1024 SIMDs x the clock measured in the same profile, GRBM_GUI_ACTIVE / 8 / kernel time); salu_issue_frac: scalar instructions / (256 CUs x that clock): a CU has one scalar unit for its four SIMDs"}
        except Exception as e:      # a malformed profile file must not fail the bench
            roofline["limiters_error"] = str(e)

    out = {
        "metric": "Mrays/s (all ray types) on Bistro-like 1080p 64spp; s/frame = ms_per_step/1000",
        "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(sec_per_frame * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene} (procedural stand-in, seed {args.seed}, detail {args.detail}) "
                               f"{w}x{h} {spp}spp PathTracer<{args.bounces}> env-map IS + sun + MIS, tile-interleaved over {world} GPU(s)",
                   "triangles": scene.stats()["triangles"], "instances": scene.stats()["instances"],
                   "parallelism": f"tiles8x8 mod {world}" + (" + one RCCL all-gather of owned tiles per frame" if world > 1 else ""),
                   "frames_in_flight": nfl},
        "sec_per_frame": round(sec_per_frame, 4),
        "mprimary_per_s": round(primary_per_frame / sec_per_frame / 1e6, 2),
        "rays_per_frame": rays_per_frame,
        "algorithmic_bytes_per_frame": abytes_frame,                 # timed walk's own events (primary rays priced as the per-ray walk would take them)
        "frame_stream_frac": round(abytes_frame / sec_per_frame / 1e9 / HBM_PEAK_GBS, 5),              # cross-check: bytes per step / ms_per_step / 8 TB/s
        "reference_algorithmic_bytes_per_frame": abytes_frame_ref,   # the reference's walk of the same frame
        "frame_reference_stream_frac": round(abytes_frame_ref / sec_per_frame / 1e9 / HBM_PEAK_GBS, 5),   # can exceed 1: the timed kernels skip part of that walk
        "kernel_ms_per_frame": {k: round(tim[k] / args.steps, 3) for k in
                                ("raygen_ms", "packet_ms", "extend_ms", "shade_ms", "shadow_ms", "accumulate_ms", "tail_ms")},
        "kernel_ms_note": ("HIP-event time per kernel class. shadow_ms and accumulate_ms are launched on the library's side stream beside the next "
                           "iteration's k_extend (PTX_OVERLAP_SHADOW, default on) and include the time they wait for wave slots behind it: "
                           "the classes overlap and do not sum to the frame. PTX_OVERLAP_SHADOW=0 (what profiles/ are taken with) serialises them."
                           if os.environ.get("PTX_OVERLAP_SHADOW", "1") != "0" else "HIP-event time per kernel class, one stream (PTX_OVERLAP_SHADOW=0)"),
        "scene_build_s": round(build_s, 2),
        "verify": verify,
        "roofline": roofline,
    }

    if world == 1 and args.emulate_world > 1:
        out["emulated_world"] = emulate_world(pkg, loops, fbs, scene, cam, renderer, spp, args, sec_per_frame)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, scene, cam, w, h, args.bounces, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def emulate_world(pkg, loops, fbs, scene, cam, renderer, spp, args, sec_one_gpu):
    """What an N-GPU frame would cost, measured on ONE card: each rank's partition (8x8 tiles, tile % N == rank) is
    rendered in turn with everything else as in the real run.  Ranks do not talk during a frame (the exchange is one
    all-gather at the end), so the N-GPU frame time is the slowest rank's time plus the exchange.
      load balance          max / mean of the rank times
      short-queue overhead  sum of the rank times / the 1-GPU frame time (a rank's queues are N times shorter)
      adaptive frames       per-rank adaptive sample counts: how evenly the interleave spreads the noisy pixels
    Uniform frames are timed the way the bench times them: `len(loops)` contexts render alternate frames, 4 frames per
    rank, time per frame = total / 4 (with one context that is one frame at a time)."""
    n = args.emulate_world
    w, h = args.width, args.height
    cfg = pkg.AdaptiveSamplingConfig(64, 0.001) if args.adaptive_cap > spp else None      # app.rs:139-143
    times, adaptive_samples, passes = [], [], []
    loop, fb = loops[0], fbs[0]
    nfl = len(loops) if cfg is None else 1

    def one(rank_):
        if cfg is not None:
            loop.set_partition(rank_, n)
            t0 = time.perf_counter()
            r, p = pkg.render.render_frame_distributed(lambda v: int(v), loop, fb, scene, cam, renderer, spp, adaptive=cfg, max_spp=args.adaptive_cap)
            loop.sync()
            return time.perf_counter() - t0, r, p
        reps = 4
        t0 = time.perf_counter()
        for i in range(reps):
            if nfl > 1:
                loops[i % nfl].order_after_frame_tail(loops[1 - i % nfl])
            fbs[i % nfl].reset()
            fbs[i % nfl].render_passes(scene, cam, renderer, 0, spp, wait=(nfl == 1))
        for lp_k in loops[:nfl]:
            lp_k.sync()
        return (time.perf_counter() - t0) / reps, 0, 0

    try:
        full_t = None
        if cfg is not None:           # the 1-GPU reference for an adaptive frame is measured here (the bench's own frames are uniform)
            loop.set_partition(0, 1)
            one_ = lambda: (pkg.render.render_frame_distributed(lambda v: int(v), loop, fb, scene, cam, renderer, spp, adaptive=cfg, max_spp=args.adaptive_cap), loop.sync())
            one_()
            t0 = time.perf_counter(); one_(); full_t = time.perf_counter() - t0
        for r_ in range(n):
            for lp_k in loops[:nfl]:
                lp_k.set_partition(r_, n)
            one(r_)                   # warm-up: work list and path state for this partition
            dt, rendered, p = one(r_)
            times.append(dt); adaptive_samples.append(rendered); passes.append(p)
    finally:
        for lp_k in loops:
            lp_k.set_partition(0, 1)
    ref = full_t if full_t is not None else sec_one_gpu
    exch_bytes = pkg.render.tiles_bytes(w, h, n) * (n - 1)            # what one rank receives in the all-gather
    exch_s = exch_bytes / 153e9                                       # one xGMI link (point-to-point, ~153 GB/s): pessimistic, 7 links can run in parallel
    mx, mean = max(times), sum(times) / n
    return {"world": n, "frames_in_flight": nfl, "rank_s": [round(t, 4) for t in times], "max_over_mean": round(mx / mean, 4),
            "sum_over_one_gpu_frame": round(sum(times) / ref, 4), "one_gpu_frame_s": round(ref, 4),
            "exchange_bytes_received_per_rank": exch_bytes, "exchange_s_one_link": round(exch_s, 6),
            "predicted_frame_s": round(mx + exch_s, 4), "predicted_speedup": round(ref / (mx + exch_s), 3),
            "predicted_efficiency": round(ref / (mx + exch_s) / n, 4),
            "adaptive": None if cfg is None else {"cap_spp": args.adaptive_cap, "samples_per_rank": adaptive_samples, "passes_per_rank": passes,
                                                  "max_over_mean_samples": round(max(adaptive_samples) / max(1.0, sum(adaptive_samples) / n), 4)}}


def cpu_baseline(pkg, scene, cam, w, h, bounces, budget_s):
    """The C++ restatement of the reference's CPU path (oracle, libm build) on this box's host
    cores, pixel-parallel like rayon's into_par_iter, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_py import Oracle
    oracle = Oracle("libm")
    threads = max(1, min(16, os.cpu_count() or 1))   # the GPU box gives one job a 16-CPU share
    oracle.set_threads(threads)
    osc = oracle.scene(scene)
    rng = np.random.default_rng(1)

    tx_n, ty_n = (w + 7) // 8, (h + 7) // 8
    tile_order = rng.permutation(tx_n * ty_n)

    def sample(n_px):
        """n_px pixel-samples: random 8x8 tiles at sample_index 0, then the same tiles at sample_index 1, ..."""
        n_tiles = min(len(tile_order), max(1, -(-n_px // 64)))
        xs, ys = [], []
        for t in tile_order[:n_tiles]:
            tx, ty = t % tx_n, t // tx_n
            for ly in range(8):
                for lx in range(8):
                    x, y = tx * 8 + lx, ty * 8 + ly
                    if x < w and y < h:
                        xs.append(x); ys.append(y)
        xs, ys = np.array(xs), np.array(ys)
        reps = max(1, -(-n_px // len(xs)))
        return np.concatenate([np.stack([xs, ys, np.full(len(xs), k)], axis=1) for k in range(reps)]).astype(np.uint32)

    # grow the sample until one run takes about budget_s (the first, small run also warms the thread pool)
    n_px, dt, work = 64 * 64, 0.0, None
    for _ in range(4):
        work = sample(n_px)
        oracle.reset_counters()
        t0 = time.perf_counter()
        osc.render_pixels(cam, w, h, work, bounces)
        dt = time.perf_counter() - t0
        if dt >= 0.6 * budget_s:
            break
        n_px = int(len(work) * min(budget_s / max(dt, 1e-3), 64.0))
    c = oracle.counters()
    rays = c["rays_closest"] + c["rays_any"]
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{len(work)} pixel-samples (random 8x8 tiles, sample indices 0..{int(work[:, 2].max())}) of the same scene/camera/bounces, "
                      f"{dt:.1f} s; C++ restatement of the reference CPU path (no Rust toolchain here), AVX2 node test",
            "mprimary_per_s": round(len(work) / dt / 1e6, 4),
            "sec_per_frame_extrapolated": round(dt / len(work) * w * h * 64, 1)}


if __name__ == "__main__":
    main()
