"""Measurement (not collected by pytest): how long does one closest-hit / any-hit launch take when it holds only a few
rays?  The duration of a launch in the late iterations of a frame is the lifetime of its longest ray; a per-wave loop
(a "tail kernel") would replace "longest of the launch" by "longest of the wave's 64".  This prints launch time against
batch size for random subsets of the bench scene's bounce rays.   python tests/probe_tail_latency.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg._abi
w, h = 1920, 1080
sc = pkg.Scene.procedural("bistro_like", 1, 1.0, 1)
cam = sc.default_camera(w / h)
lp = pkg.HipRenderLoop(0)
prim = lp.primary_rays(sc, cam, w, h, 0)
hits = lp.trace_closest(sc, prim)
ok = hits["inst_idx"] != abi.PTX_MISS
rng = np.random.default_rng(1)
# third-bounce-like rays: bounce twice with random directions
rays = prim[ok]
t = hits["t"][ok]
for _ in range(2):
    o = rays["origin"] + rays["direction"] * t[:, None]
    d = rng.normal(size=o.shape).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    nxt = np.zeros(len(o), dtype=abi.ray_dtype())
    nxt["origin"], nxt["direction"], nxt["t_max"] = o + d * np.float32(1e-3), d, np.inf
    hh = lp.trace_closest(sc, nxt)
    keep = hh["inst_idx"] != abi.PTX_MISS
    rays, t = nxt[keep], hh["t"][keep]
bounce = rays
print(f"{len(bounce)} rays that start on a surface after two bounces", flush=True)
for mode in ("closest", "any"):
    fn = lp.trace_closest if mode == "closest" else lp.trace_any
    for n in (64, 512, 4096, 32768, 262144, len(bounce)):
        ms = []
        for rep in range(6):
            sub = bounce[rng.choice(len(bounce), n, replace=False)] if n < len(bounce) else bounce
            fn(sc, sub)
            lp.set_profiling(True); lp.reset_timings()
            fn(sc, sub)
            tm = lp.timings()
            lp.set_profiling(False)
            ms.append(tm["extend_ms"] if mode == "closest" else tm["shadow_ms"])
        ms = np.array(ms)
        print(f"{mode:8s} {n:8d} rays: launch {ms.min() * 1e3:8.1f} .. {np.median(ms) * 1e3:8.1f} .. {ms.max() * 1e3:8.1f} us (min / median / max of 6 subsets)", flush=True)
