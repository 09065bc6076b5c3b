import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("PT_VARIANT_LIB"):
    pkg.render.DEVICE_LIB_PATH = os.path.abspath(os.environ["PT_VARIANT_LIB"])
w, h = 640, 360
scene = pkg.Scene.procedural("cornell", 1, 1.0, 0)
cam = scene.default_camera(w / h)
loop = pkg.HipRenderLoop(0)
r = pkg.PathTracer(8)
for k in range(4):
    loop.set_profiling(k >= 2); loop.reset_timings()
    t0 = time.perf_counter(); loop.render_frame(scene, cam, r, w, h, 16); dt = time.perf_counter() - t0
    print("frame %.2f ms" % (dt * 1e3), {a: round(b, 3) for a, b in loop.timings().items()} if k >= 2 else "", flush=True)
