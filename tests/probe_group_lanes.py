"""Measurement (not collected by pytest): closest-hit with G lanes per ray (PTX_GROUP_LANES=4|8, pt_traverse_group.h)
against the per-ray kernels, on primary and bounce ray batches of a BASELINE scene; hit records must be identical.
    python tests/probe_group_lanes.py [scene] [detail] [flags]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg._abi
name = sys.argv[1] if len(sys.argv) > 1 else "bistro_like"
detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w, h = 1920, 1080
sc = pkg.Scene.procedural(name, 1, detail, flags)
cam = sc.default_camera(w / h)


def make(env):
    for k in ("PTX_GROUP_LANES", "PTX_EXACT_ORDER", "PTX_NO_FORWARD", "PTX_WIDE_RAY_BVH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    lp = pkg.HipRenderLoop(0)
    lp._bind(sc, cam)
    return lp


base = make({})
prim = base.primary_rays(sc, cam, w, h, 0)
hits = base.trace_closest(sc, prim)
ok = hits["inst_idx"] != abi.PTX_MISS
rng = np.random.default_rng(1)
o = prim["origin"][ok] + prim["direction"][ok] * hits["t"][ok, None]
d = rng.normal(size=o.shape).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
bounce = np.zeros(len(o), dtype=abi.ray_dtype())
bounce["origin"], bounce["direction"], bounce["t_max"] = o + d * np.float32(1e-3), d, np.inf
print(f"{name} {sc.stats()} primary {len(prim)} hitfrac {ok.mean():.3f} bounce {len(bounce)}", flush=True)
want = {"primary": hits, "bounce": base.trace_closest(sc, bounce)}
for label, env in (("per-ray nearest-first (shipped)", {}), ("per-ray reference order + forwarding", {"PTX_EXACT_ORDER": "1"}),
                   ("4 lanes per ray, 4-wide tree", {"PTX_GROUP_LANES": "4"}), ("8 lanes per ray, 8-wide tree", {"PTX_GROUP_LANES": "8"})):
    lp = make(env)
    for rl, rays in (("primary", prim), ("bounce", bounce)):
        got = lp.trace_closest(sc, rays)
        same = np.array_equal(got.view(np.uint8), want[rl].view(np.uint8))
        lp.set_profiling(True); lp.reset_timings()
        for _ in range(3):
            lp.trace_closest(sc, rays)
        t = lp.timings()
        lp.set_profiling(False)
        ms = t["extend_ms"] / 3
        print(f"{label:40s} {rl:8s} {ms:8.3f} ms {len(rays) / ms / 1e3:8.1f} Mrays/s  identical {same}", flush=True)
    lp.close()
