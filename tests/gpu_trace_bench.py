"""Traversal micro-benchmark (not collected by pytest): k_trace_batch on primary and bounce rays of the bench scene.
Prints kernel-only time (HIP events on the library stream) and per-ray event counts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg._abi
if os.environ.get("PT_VARIANT_LIB"):      # experiments: a library built by tools/build_variant.sh
    pkg.render.DEVICE_LIB_PATH = os.path.abspath(os.environ["PT_VARIANT_LIB"])
name =sys.argv[1] if len(sys.argv) > 1 else "bistro_like"
detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
w, h = 1920, 1080
sc = pkg.Scene.procedural(name, 1, detail, 1)
cam = sc.default_camera(w / h)
loop = pkg.HipRenderLoop(0)
prim = loop.primary_rays(sc, cam, w, h, 0)
hits = loop.trace_closest(sc, prim)
ok = hits["inst_idx"] != abi.PTX_MISS
rng = np.random.default_rng(1)
o = prim["origin"][ok] + prim["direction"][ok] * hits["t"][ok, None]
d = rng.normal(size=o.shape).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
bounce = np.zeros(len(o), dtype=abi.ray_dtype())
bounce["origin"], bounce["direction"], bounce["t_max"] = o + d * np.float32(1e-3), d, np.inf
print(f"{name} {sc.stats()} primary {len(prim)} hitfrac {ok.mean():.3f} bounce {len(bounce)}", flush=True)

for label, rays in (("primary", prim), ("bounce", bounce)):
    for mode in ("closest", "any"):
        fn = loop.trace_closest if mode == "closest" else loop.trace_any
        loop.set_counting(True); loop.reset_counters()
        fn(sc, rays)
        c = loop.counters()
        loop.set_counting(False)
        fn(sc, rays)
        loop.set_profiling(True); loop.reset_timings()
        reps = 3
        for _ in range(reps):
            fn(sc, rays)
        t = loop.timings()
        loop.set_profiling(False)
        ms = (t["extend_ms"] if mode == "closest" else t["shadow_ms"]) / reps
        n = len(rays)
        b = pkg._abi.algorithmic_bytes(c)
        print(f"{label:8s} {mode:8s} {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrays/s  alg {b / ms / 1e6:7.1f} GB/s | per ray: nodes {c['node_visits'] / n:6.1f} "
              f"children {c['node_children'] / n:6.1f} tris {c['triangle_tests'] / n:5.1f} inst_enter {c['instance_entries'] / n:5.1f} inst_hit {c['instance_hits'] / n:4.1f}", flush=True)
