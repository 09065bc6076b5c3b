"""C5 anatomy: wall time, launches and per-class kernel time of adaptive passes on the SSS bistro scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("PT_VARIANT_LIB"):
    pkg.render.DEVICE_LIB_PATH = os.path.abspath(os.environ["PT_VARIANT_LIB"])
w, h = 1920, 1080
scene = pkg.Scene.procedural("bistro_like", 1, 1.0, 3)
cam = scene.default_camera(w / h)
loop = pkg.HipRenderLoop(0)
r = pkg.PathTracer(16)
fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
fb.reset()
t0 = time.perf_counter(); fb.render_passes(scene, cam, r, 0, 64); loop.sync(); print("64 uniform passes %.3f s" % (time.perf_counter() - t0), flush=True)
cfg = pkg.AdaptiveSamplingConfig(64, 0.001)
for k in range(6):
    prof = k >= 3
    loop.set_profiling(prof); loop.reset_timings()
    t0 = time.perf_counter(); n = loop.render_adaptive_pass(scene, cam, r, fb, cfg); loop.sync(); dt = time.perf_counter() - t0
    print("pass %d: %d samples, %.2f ms wall, profiling=%s %s" % (k, n, dt * 1e3, prof, {a: round(b, 2) for a, b in loop.timings().items()} if prof else ""), flush=True)
