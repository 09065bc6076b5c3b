#!/usr/bin/env python3
"""Fill BASELINE.md §4: every config of BASELINE.json on one MI355X (through ptx_render_frame) next to the CPU port
(bench.cpu_baseline: the oracle's libm build on the box's host cores, bounded sample).  Run on the GPU box:
    python tools/measure_configs.py > gpurun_out/configs.md"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

CONFIGS = [
    # name, scene, detail, flags, w, h, spp, bounces, adaptive (min, thr, cap) or None
    ("C1 Cornell, --benchmark defaults", "cornell", 1.0, 0, 640, 360, 16, 8, None),
    ("C1 helmet-like, --benchmark defaults", "helmet_like", 1.0, 0, 640, 360, 16, 8, None),
    ("C2 Sponza-like, env-map IS + MIS + sun", "sponza_like", 1.0, 1, 1920, 1080, 64, 16, None),
    ("C3 Bistro-like, env-map IS + MIS + sun", "bistro_like", 1.0, 1, 1920, 1080, 64, 16, None),
    ("C4 10k-instance TLAS stress", "instanced_stress", 1.0, 0, 3840, 2160, 32, 4, None),
    ("C5 Bistro-like adaptive + SSS", "bistro_like", 1.0, 3, 1920, 1080, 64, 16, (64, 0.001, 256)),
]


def main():
    pkg = ge.load_package()
    if os.environ.get("PT_VARIANT_LIB"):      # kernel experiments only: a library built by tools/build_variant.sh
        pkg.render.DEVICE_LIB_PATH = os.path.abspath(os.environ["PT_VARIANT_LIB"])
    cpu_s = float(os.environ.get("CPU_SECONDS", "8"))
    rows = []
    only = os.environ.get("ONLY")          # e.g. ONLY=C5: one row
    for name, scene_name, detail, flags, w, h, spp, bounces, adaptive in CONFIGS:
        if only and not name.startswith(only):
            continue
        t0 = time.time()
        scene = pkg.Scene.procedural(scene_name, 1, detail, flags)
        build_s = time.time() - t0
        cam = scene.default_camera(w / h)
        loop = pkg.HipRenderLoop(0)
        renderer = pkg.PathTracer(bounces)
        cfg = pkg.AdaptiveSamplingConfig(adaptive[0], adaptive[1]) if adaptive else None
        cap = adaptive[2] if adaptive else 0

        def frame():
            return loop.render_frame(scene, cam, renderer, w, h, spp, adaptive=cfg, max_spp=cap)

        loop.set_counting(True); loop.reset_counters()
        _, _, stats = frame()
        cnt = loop.counters()
        loop.set_counting(False)
        frame()                                   # warm-up (allocations, caches)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            frame()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rays = cnt["rays_closest"] + cnt["rays_any"]
        primaries = stats["total_samples"]
        ab = pkg._abi.algorithmic_bytes(cnt, texel_bytes_per_lookup=48, adaptive=adaptive is not None)
        cpu = bench.cpu_baseline(pkg, scene, cam, w, h, bounces, cpu_s) if cpu_s > 0 else {"cores": 0, "mprimary_per_s": 1e-9, "value": 0.0}
        rows.append({
            "config": name, "res_spp": f"{w}x{h} / {spp}" + (f" + adaptive to {cap}" if adaptive else ""),
            "triangles": scene.stats()["triangles"], "instances": scene.stats()["instances"], "scene_build_s": round(build_s, 2),
            "gpu_s_per_frame": round(best, 4), "gpu_mprimary_s": round(primaries / best / 1e6, 2), "gpu_mrays_s": round(rays / best / 1e6, 1),
            "rays_per_primary": round(rays / max(primaries, 1), 2), "alg_bytes_per_ray": round(ab / max(rays, 1)),
            "hbm_frac": round(ab / best / 8e12, 3), "adaptive_passes": stats["adaptive_passes"], "total_samples": primaries,
            "cpu_cores": cpu["cores"], "cpu_mprimary_s": cpu["mprimary_per_s"], "cpu_mrays_s": cpu["value"],
            "cpu_s_per_frame_extrapolated": round(primaries / (cpu["mprimary_per_s"] * 1e6), 1),
        })
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
        del loop, scene
    print("| config | resolution / spp | CPU port (cores) s/frame (extrapolated) | CPU Mprimary/s | 1×MI355X s/frame | Mprimary/s | Mrays/s (all) | algorithmic B/ray | reference byte stream ÷ 8 TB/s (NOT HBM utilisation) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['config']} ({r['triangles']:,} tris, {r['instances']} inst) | {r['res_spp']} | {r['cpu_s_per_frame_extrapolated']} ({r['cpu_cores']}) | "
              f"{r['cpu_mprimary_s']} | {r['gpu_s_per_frame']} | {r['gpu_mprimary_s']} | {r['gpu_mrays_s']} | {r['alg_bytes_per_ray']} | {100 * r['hbm_frac']:.1f} |")
    print()
    print("```json")
    print(json.dumps(rows, indent=1))
    print("```")


if __name__ == "__main__":
    main()
