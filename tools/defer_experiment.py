#!/usr/bin/env python3
"""Measurement only: what a 'search the instance with the nearest box first, then walk in reference order under its
bound' scheme could save on the bench scene.  Run on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package()
    abi = pkg._abi
    scene = pkg.Scene.procedural("bistro_like", 1, 1.0, 1)
    d = scene.desc.contents
    n_inst = d.n_instances
    lo = np.zeros((n_inst, 3)); hi = np.zeros((n_inst, 3))
    for i in range(n_inst):
        inst = d.instances[i]
        m = np.array(inst.transform[:], dtype=np.float64).reshape(4, 4).T
        mesh = d.meshes[inst.blas_index]
        bmin, bmax = np.array(mesh.bounds_min[:]), np.array(mesh.bounds_max[:])
        corners = np.array([[x, y, z, 1.0] for x in (bmin[0], bmax[0]) for y in (bmin[1], bmax[1]) for z in (bmin[2], bmax[2])])
        wc = corners @ m.T
        lo[i], hi[i] = wc[:, :3].min(0), wc[:, :3].max(0)
    w, h = 320, 180
    cam = scene.default_camera(w / h)
    os.environ["PTX_COUNT_FORWARD"] = "1"
    fw = pkg.HipRenderLoop(0)
    prim = fw.primary_rays(scene, cam, w, h, 0)
    hp = fw.trace_closest(scene, prim)
    ok = hp["inst_idx"] != abi.PTX_MISS
    rng = np.random.default_rng(1)
    o = prim["origin"][ok] + prim["direction"][ok] * hp["t"][ok, None]
    dr = rng.normal(size=(len(o), 3)).astype(np.float32)
    dr /= np.linalg.norm(dr, axis=1, keepdims=True).astype(np.float32)
    rays = np.zeros(len(o), dtype=abi.ray_dtype())
    rays["origin"], rays["direction"], rays["t_max"] = o + dr * np.float32(1e-3), dr, np.inf
    hits = fw.trace_closest(scene, rays)
    # instance boxes each ray pierces (t_enter), nearest one
    O = rays["origin"].astype(np.float64); D = rays["direction"].astype(np.float64)
    n = len(rays)
    nearest = np.full(n, -1); pierced = np.zeros(n, dtype=np.int64); before_hit = np.zeros(n, dtype=np.int64)
    t_hit = np.where(hits["inst_idx"] != abi.PTX_MISS, hits["t"], np.inf).astype(np.float64)
    for s in range(0, n, 2048):
        e = min(n, s + 2048)
        inv = 1.0 / D[s:e, None, :]
        t1 = (lo[None] - O[s:e, None, :]) * inv; t2 = (hi[None] - O[s:e, None, :]) * inv
        tn = np.minimum(t1, t2).max(2); tf = np.maximum(t1, t2).min(2)
        tn = np.maximum(tn, 1e-3)
        hit = tn <= tf
        pierced[s:e] = hit.sum(1)
        before_hit[s:e] = (hit & (tn < t_hit[s:e, None])).sum(1)
        tn_m = np.where(hit, tn, np.inf)
        nearest[s:e] = np.where(hit.any(1), tn_m.argmin(1), -1)
    win = hits["inst_idx"].astype(np.int64)
    same = (win == nearest) & (hits["inst_idx"] != abi.PTX_MISS)
    print(f"rays {n}: instance boxes pierced (infinite ray) {pierced.mean():.1f}, entered before the hit distance {before_hit.mean():.1f}, "
          f"winner is the nearest-box instance for {same.mean():.3f} of rays (hit fraction {(hits['inst_idx'] != abi.PTX_MISS).mean():.3f})")

    def count(r, label):
        fw.set_counting(True); fw.reset_counters()
        fw.trace_closest(scene, r)
        c = fw.counters(); fw.set_counting(False)
        k = max(1, c["rays_closest"])
        print(f"{label:52s} node visits {c['node_visits'] / k:6.1f}  tri tests {c['triangle_tests'] / k:6.1f}  instance entries {c['instance_entries'] / k:5.1f}")
    count(rays, "forwarded bound (shipped)")
    tight = rays.copy()
    tight["t_max"][same] = hits["t"][same] * np.float32(1.03) + np.float32(1e-3)
    count(tight, "+ bound from the nearest-box instance when it wins")
    tight["t_max"] = np.where(np.isfinite(t_hit), t_hit * 1.03 + 1e-3, np.inf).astype(np.float32)
    count(tight, "+ perfect bound")


if __name__ == "__main__":
    main()
