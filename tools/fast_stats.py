#!/usr/bin/env python3
"""Measurement only: how many rays of a frame the nearest-first kernel hands to the reference-order kernel.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "bistro_like"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scene = pkg.Scene.procedural(name, 1, 1.0, flags)
w, h, spp = 1920, 1080, 16
cam = scene.default_camera(w / h)
loop = pkg.HipRenderLoop(0)
fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
loop.reset_counters()
loop.set_profiling(True); loop.reset_timings()
fb.render_passes(scene, cam, pkg.PathTracer(16), 0, spp)
loop.sync()
t = loop.timings(); d = loop.debug_stats()
print(name, "fast_order", d["fast_order"], "second attempts", d["fast_redo_rays"], {k: round(v, 2) for k, v in t.items() if k.endswith("_ms")})
loop.set_profiling(False)
loop.set_counting(True); loop.reset_counters()
fb.reset(); fb.render_passes(scene, cam, pkg.PathTracer(16), 0, spp); loop.sync()
c = loop.counters(); loop.set_counting(False)
print("closest rays", c["rays_closest"], "primary", w * h * spp, "redo fraction of non-primary closest rays %.4f" % (d["fast_redo_rays"] / max(1, c["rays_closest"] - w * h * spp)))
