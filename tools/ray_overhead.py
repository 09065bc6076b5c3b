#!/usr/bin/env python3
"""Measurement only: the per-ray fixed cost of the per-ray closest-hit kernel (queue claim, ray fetch, Ray::new, the root
visit, the result write) — rays that leave the scene at once — beside real bounce rays.  Run on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package()
    abi = pkg._abi
    scene = pkg.Scene.procedural(sys.argv[1] if len(sys.argv) > 1 else "bistro_like", 1, 1.0, 1)
    loop = pkg.HipRenderLoop(0)
    n = 1 << 24
    rng = np.random.default_rng(3)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:, 1] = np.abs(d[:, 1]) + np.float32(0.2)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    miss = np.zeros(n, dtype=abi.ray_dtype())
    miss["origin"] = (rng.uniform(-50, 50, size=(n, 3)) + np.array([0.0, 5000.0, 0.0])).astype(np.float32)   # far above everything, going up
    miss["direction"], miss["t_max"] = d, np.inf

    def timed(rays, label, any_hit=False):
        (loop.trace_any if any_hit else loop.trace_closest)(scene, rays[:1024])
        loop.set_profiling(True); loop.reset_timings()
        out = (loop.trace_any if any_hit else loop.trace_closest)(scene, rays)
        t = loop.timings(); loop.set_profiling(False)
        ms = t["shadow_ms"] if any_hit else t["extend_ms"]
        hit = float(out.mean()) if any_hit else float((out["inst_idx"] != abi.PTX_MISS).mean())
        print(f"{label:46s} {len(rays) / 1e6:6.1f} M rays  {ms:8.3f} ms  {ms * 1e6 / len(rays):6.3f} ns/ray  hit fraction {hit:.3f}")
        return out
    timed(miss, "rays that miss the root (closest)")
    timed(miss, "rays that miss the root (any-hit)", True)
    # real bounce rays: from the hit points of primary rays
    w, h = 1920, 1080
    cam = scene.default_camera(w / h)
    prim = np.concatenate([loop.primary_rays(scene, cam, w, h, s) for s in range(4)])
    hp = timed(prim, "primary rays, per-ray kernel")
    ok = hp["inst_idx"] != abi.PTX_MISS
    o = prim["origin"][ok] + prim["direction"][ok] * hp["t"][ok, None]
    dd = rng.normal(size=(len(o), 3)).astype(np.float32)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True).astype(np.float32)
    b = np.zeros(len(o), dtype=abi.ray_dtype())
    b["origin"], b["direction"], b["t_max"] = o + dd * np.float32(1e-3), dd, np.inf
    timed(b, "bounce rays (closest)")
    timed(b, "bounce rays (any-hit)", True)


if __name__ == "__main__":
    main()
