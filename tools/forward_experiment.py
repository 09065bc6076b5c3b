#!/usr/bin/env python3
"""Measurement only: traversal events per bounce ray on the bench scene (a) as the reference walks (counted kernels),
(b) with the forwarded bound, (c) with the forwarded bound and t_max = 1.03 x the true hit distance (what a perfect
first guess of the hit would leave for the reference-order walk).  Run on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package()
    abi = pkg._abi
    scene = pkg.Scene.procedural("bistro_like", 1, 1.0, 1)
    w, h = 480, 270
    cam = scene.default_camera(w / h)
    ref = pkg.HipRenderLoop(0)
    prim = ref.primary_rays(scene, cam, w, h, 0)
    hp = ref.trace_closest(scene, prim)
    ok = hp["inst_idx"] != abi.PTX_MISS
    rng = np.random.default_rng(1)
    o = prim["origin"][ok] + prim["direction"][ok] * hp["t"][ok, None]
    d = rng.normal(size=(len(o), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays = np.zeros(len(o), dtype=abi.ray_dtype())
    rays["origin"], rays["direction"], rays["t_max"] = o + d * np.float32(1e-3), d, np.inf
    hits = ref.trace_closest(scene, rays)

    def count(loop, r, label):
        loop.set_counting(True)
        loop.reset_counters()
        g = loop.trace_closest(scene, r)
        c = loop.counters()
        dbg = loop.debug_stats()
        loop.set_counting(False)
        n = max(1, c["rays_closest"])
        print(f"{label:48s} node visits {c['node_visits'] / n:6.1f}  tri tests {c['triangle_tests'] / n:6.1f}  instance entries {c['instance_entries'] / n:5.1f}"
              f"  instance hits {c['instance_hits'] / n:5.2f}")
        v = dbg["vote_closest"]
        names = ("NODE", "TRI", "ENTER", "EXIT")
        print("    vote: " + ", ".join(f"{names[k]} runs in {v[1 + k] / max(1, v[0]):.2f} of iterations with {v[5 + k] / max(1, v[1 + k]):.1f} lanes" for k in range(4)) +
              f"; {v[0] / n:.1f} wave iterations per ray x 64")
        return g
    count(ref, rays, "(a) reference walk")
    os.environ["PTX_COUNT_FORWARD"] = "1"
    fw = pkg.HipRenderLoop(0)
    g = count(fw, rays, "(b) forwarded bound")
    assert np.array_equal(g.view(np.uint8), hits.view(np.uint8))
    tight = rays.copy()
    hit = hits["inst_idx"] != abi.PTX_MISS
    tight["t_max"][hit] = hits["t"][hit] * np.float32(1.03) + np.float32(1e-3)
    g = count(fw, tight, "(c) forwarded + t_max = 1.03 t_hit")
    assert np.array_equal(g.view(np.uint8), hits.view(np.uint8))
    print(f"rays {len(rays)}, hit fraction {hit.mean():.3f}")


if __name__ == "__main__":
    main()
