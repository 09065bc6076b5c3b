"""Ad-hoc GPU diagnostics (not collected by pytest): prints mismatch statistics per scene."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import __graft_entry__ as ge  # noqa: E402
from oracle_py import Oracle  # noqa: E402

pkg = ge.load_package()
od = Oracle("det")
loop = pkg.HipRenderLoop(0)


def probe(name, flags, detail, w, h, bounces, spp=2):
    sc = pkg.Scene.procedural(name, 1, detail, flags)
    cam = sc.default_camera(w / h)
    osc = od.scene(sc)
    rays_c = osc.primary_rays(cam, w, h, 0)
    rays_g = loop.primary_rays(sc, cam, w, h, 0)
    same_rays = np.array_equal(rays_c.view(np.uint8), rays_g.view(np.uint8))
    hc = osc.trace_closest(rays_c)
    t0 = time.time()
    hg = loop.trace_closest(sc, rays_c)
    tg = time.time() - t0
    ids = (hc["inst_idx"] == hg["inst_idx"]) & (hc["tri_idx"] == hg["tri_idx"])
    tb = hc["t"].view(np.uint32) == hg["t"].view(np.uint32)
    uvb = (hc["u"].view(np.uint32) == hg["u"].view(np.uint32)) & (hc["v"].view(np.uint32) == hg["v"].view(np.uint32))
    ac = osc.trace_any(rays_c)
    ag = loop.trace_any(sc, rays_c)
    print(f"[{name} f{flags} d{detail}] {sc.stats()} rays_equal={same_rays} ids={ids.mean():.6f} t={tb.mean():.6f} uv={uvb.mean():.6f} "
          f"any={np.mean(ac == ag):.6f} hitfrac={np.mean(hc['inst_idx'] != 0xFFFFFFFF):.3f} gpu_trace_s={tg:.3f}", flush=True)
    for s in range(spp):
        ic = osc.render_pass(cam, w, h, s, bounces)
        t0 = time.time()
        ig = loop.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, s)
        tg = time.time() - t0
        eq = (ic.view(np.uint32) == ig.view(np.uint32)).all(axis=2)
        d = np.abs(ic - ig)
        print(f"   pass {s}: pixels bit-equal {eq.mean():.6f} max|d| {d.max():.3e} mean_c {ic.mean():.6f} mean_g {ig.mean():.6f} gpu_s={tg:.3f}", flush=True)
        if eq.mean() < 1.0:
            bad = np.argwhere(~eq)[:5]
            for y, x in bad:
                print("      bad", x, y, ic[y, x], ig[y, x], flush=True)


if __name__ == "__main__":
    probe("cornell", 0, 1.0, 64, 36, 8)
    probe("cornell", 1, 1.0, 96, 54, 8)
    probe("helmet_like", 0, 0.05, 96, 54, 8)
    probe("sponza_like", 1, 0.05, 96, 54, 16)
    probe("instanced_stress", 0, 0.05, 96, 54, 4)
    probe("glass_test", 1, 0.3, 96, 54, 16)
    probe("sss_test", 1, 0.3, 96, 54, 16)
    probe("bistro_like", 3, 0.02, 96, 54, 16)
