"""C5: time the adaptive phase with PTX_ADAPTIVE_BATCH from argv."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
w, h = 1920, 1080
scene = pkg.Scene.procedural("bistro_like", 1, 1.0, 3)
cam = scene.default_camera(w / h)
r = pkg.PathTracer(16)
cfg = pkg.AdaptiveSamplingConfig(64, 0.001)
for batch in [int(a) for a in sys.argv[1:]]:
    os.environ["PTX_ADAPTIVE_BATCH"] = str(batch)
    loop = pkg.HipRenderLoop(0)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    fb.reset()
    t0 = time.perf_counter(); fb.render_passes(scene, cam, r, 0, 64); loop.sync(); tu = time.perf_counter() - t0
    t0 = time.perf_counter(); p, n = loop.render_adaptive_passes(scene, cam, r, fb, cfg, 192); loop.sync(); ta = time.perf_counter() - t0
    s = fb.read()
    print("batch %3d: uniform %.3f s, adaptive %.3f s (%d passes, %d samples), checksum %r %d" % (batch, tu, ta, p, n, float(s[0].astype('f8').sum()), int(s[2].sum())), flush=True)
    del fb, loop
