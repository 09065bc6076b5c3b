"""Measurement (not collected by pytest): what would reordering the bounce rays of an iteration buy?  Traces the same
second-bounce rays of the bench scene in several queue orders and prints the closest-hit launch time of each.
    python tests/probe_ray_sort.py"""
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
rng = np.random.default_rng(1)
parts = []
for s in range(4):                       # 4 samples per pixel, sample-major like a batch of passes
    prim = lp.primary_rays(sc, cam, w, h, s)
    hits = lp.trace_closest(sc, prim)
    ok = hits["inst_idx"] != abi.PTX_MISS
    o = prim["origin"][ok] + prim["direction"][ok] * hits["t"][ok][:, None]
    d = rng.normal(size=o.shape).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    nxt = np.zeros(len(o), dtype=abi.ray_dtype())
    nxt["origin"], nxt["direction"], nxt["t_max"] = o + d * np.float32(1e-3), d, np.inf
    parts.append(nxt)
rays = np.concatenate(parts)
n = len(rays)
print(f"{n} bounce rays (pixel order within each of 4 samples)", flush=True)


def morton(o, bits):
    lo, hi = o.min(axis=0), o.max(axis=0)
    q = np.minimum(((o - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.uint32), (1 << bits) - 1)
    code = np.zeros(len(o), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1).astype(np.uint64) << np.uint64(3 * b + a)
    return code


octant = ((rays["direction"][:, 0] < 0).astype(np.uint64) | ((rays["direction"][:, 1] < 0).astype(np.uint64) << np.uint64(1)) |
          ((rays["direction"][:, 2] < 0).astype(np.uint64) << np.uint64(2)))
orders = {
    "as generated (pixel order)": np.arange(n),
    "shuffled": rng.permutation(n),
    "octant (stable)": np.argsort(octant, kind="stable"),
    "morton5(origin)": np.argsort(morton(rays["origin"], 5), kind="stable"),
    "morton5(origin) then octant": np.argsort((morton(rays["origin"], 5) << np.uint64(3)) | octant, kind="stable"),
    "octant then morton5(origin)": np.argsort((octant << np.uint64(15)) | morton(rays["origin"], 5), kind="stable"),
    "morton8(origin) then octant": np.argsort((morton(rays["origin"], 8) << np.uint64(3)) | octant, kind="stable"),
    "octant then morton8(origin)": np.argsort((octant << np.uint64(24)) | morton(rays["origin"], 8), kind="stable"),
}
want = None
for name, perm in orders.items():
    sub = np.ascontiguousarray(rays[perm])
    got = lp.trace_closest(sc, sub)
    ms = []
    for rep in range(3):
        lp.set_profiling(True); lp.reset_timings()
        lp.trace_closest(sc, sub)
        ms.append(lp.timings()["extend_ms"])
        lp.set_profiling(False)
    back = np.empty_like(got); back[perm] = got
    if want is None:
        want = back
    assert np.array_equal(back.view(np.uint8), want.view(np.uint8)), name      # the order changes no hit
    print(f"{name:32s} closest-hit launch {min(ms):8.2f} ms  ({n / min(ms) / 1e3:7.1f} Mrays/s)", flush=True)
