#!/usr/bin/env python3
"""Per-kernel-class device time of one frame of a BASELINE config (HIP events on the library stream).
    python tools/frame_breakdown.py C5 [uniform|adaptive|both]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from tools.measure_configs import CONFIGS  # noqa: E402


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "C5"
    part = sys.argv[2] if len(sys.argv) > 2 else "both"
    pkg = ge.load_package()
    if os.environ.get("PT_VARIANT_LIB"):      # kernel experiments only: a library built by tools/build_variant.sh
        pkg.render.DEVICE_LIB_PATH = os.path.abspath(os.environ["PT_VARIANT_LIB"])
    for name, scene_name, detail, flags, w, h, spp, bounces, adaptive in CONFIGS:
        if not name.startswith(want):
            continue
        scene = pkg.Scene.procedural(scene_name, 1, detail, flags)
        cam = scene.default_camera(w / h)
        loop = pkg.HipRenderLoop(0)
        if os.environ.get("PT_PARTITION"):        # "rank,world": this rank's tiles only (what one GPU of N renders)
            rank, world = (int(x) for x in os.environ["PT_PARTITION"].split(","))
            loop.set_partition(rank, world)
        r = pkg.PathTracer(bounces)
        fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
        cfg = pkg.AdaptiveSamplingConfig(adaptive[0], adaptive[1]) if adaptive else None

        def uniform():
            fb.reset()
            fb.render_passes(scene, cam, r, 0, spp)
            loop.sync()

        def adapt():
            out = loop.render_adaptive_passes(scene, cam, r, fb, cfg, adaptive[2] - spp) if adaptive else (0, 0)
            loop.sync()
            return out

        uniform(); adapt()                # warm-up
        t0 = time.perf_counter(); uniform(); print(json.dumps({"config": name, "unprofiled_uniform_wall_s": round(time.perf_counter() - t0, 4)}), flush=True)
        for label, fn in (("uniform", uniform), ("adaptive", adapt)):
            if label == "uniform" or (adaptive and part != "uniform"):
                if label == "adaptive" and part == "adaptive":
                    pass
                loop.set_profiling(True); loop.reset_timings()
                t0 = time.perf_counter()
                res = fn()
                dt = time.perf_counter() - t0
                tim = loop.timings()
                loop.set_profiling(False)
                print(json.dumps({"config": name, "part": label, "wall_s": round(dt, 4), "result": res,
                                  "timings": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tim.items()}}), flush=True)


if __name__ == "__main__":
    main()
