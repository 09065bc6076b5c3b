import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
w, h = 1920, 1080
scene = pkg.Scene.procedural("bistro_like", 1, 1.0, 3)
cam = scene.default_camera(w / h)
r = pkg.PathTracer(16)
loop = pkg.HipRenderLoop(0)
fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
for k in range(2):
    fb.reset(); loop.set_profiling(k == 1); loop.reset_timings()
    t0 = time.perf_counter(); fb.render_passes(scene, cam, r, 0, 64); loop.sync(); tu = time.perf_counter() - t0
    print("uniform %.3f s" % tu, {a: round(b, 2) for a, b in loop.timings().items()}, flush=True)
