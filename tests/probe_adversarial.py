"""Measurement, not a test (pytest does not collect it): how often do the walk modes of the closest-hit kernels
differ from the det oracle on the adversarial scenes of tests/adversarial_scenes.py?
    python tests/probe_adversarial.py [out.json]
Modes: default (forwarded bound + nearest-first), PTX_EXACT_ORDER=1 (forwarded bound only), PTX_EXACT_ORDER=1
PTX_NO_FORWARD=1 (the reference's walk operation for operation), PTX_PACKET_BATCH=1 (packets)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import __graft_entry__ as ge  # noqa: E402
import adversarial_scenes as adv  # noqa: E402
from oracle_py import Oracle  # noqa: E402

pkg = ge.load_package()
abi = pkg._abi
oracle = Oracle("det")

MODES = {
    "reference_walk": {"PTX_EXACT_ORDER": "1", "PTX_NO_FORWARD": "1"},
    "no_cert": {"PTX_NO_SHAPE_CERT": "1"},
    "no_cert_forward_only": {"PTX_NO_SHAPE_CERT": "1", "PTX_EXACT_ORDER": "1"},
}


def make_loop(env):
    for k in ("PTX_EXACT_ORDER", "PTX_NO_FORWARD", "PTX_PACKET_BATCH", "PTX_NO_SHAPE_CERT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    lp = pkg.HipRenderLoop(0)
    return lp


def run(name, sc, rays, out, labels=None):
    osc = oracle.scene(sc)
    t0 = time.time()
    want = osc.trace_closest(rays)
    want_any = osc.trace_any(rays)
    t_or = time.time() - t0
    hit = want["inst_idx"] != abi.PTX_MISS
    row = {"rays": int(len(rays)), "hit_share": float(hit.mean()), "oracle_s": round(t_or, 2)}
    for mode, env in MODES.items():
        lp = make_loop(env)
        lp.reset_counters()
        got = lp.trace_closest(sc, rays)
        dbg = lp.debug_stats()
        bad = adv.diff_rows(got, want)
        ids = np.flatnonzero((got["inst_idx"] != want["inst_idx"]) | (got["tri_idx"] != want["tri_idx"]))
        any_bad = int((lp.trace_any(sc, rays) != want_any).sum())
        row[mode] = {"mismatch": int(len(bad)), "id_mismatch": int(len(ids)), "any_mismatch": any_bad, "fast_order": dbg.get("fast_order"),
                     "forwarding": dbg.get("bound_forwarding"), "redo_share": dbg.get("fast_redo_rays", 0) / max(1, len(rays)),
                     "uncertified_meshes": dbg.get("uncertified_meshes")}
        if labels is not None and len(bad):
            row[mode]["by_class"] = {str(c): int((labels[bad] == c).sum()) for c in np.unique(labels[bad])}
        if len(bad):
            k = bad[0]
            row[mode]["first"] = {"ray": [float(x) for x in list(rays[k]["origin"]) + list(rays[k]["direction"])] + [float(rays[k]["t_max"])],
                                  "got": [int(got[k]["inst_idx"]), int(got[k]["tri_idx"]), float(got[k]["t"])],
                                  "want": [int(want[k]["inst_idx"]), int(want[k]["tri_idx"]), float(want[k]["t"])]}
        lp.close()
    # the undershoot the counted kernels see on this batch (reference walk's hits)
    lp = make_loop({})
    lp.set_counting(True)
    lp.reset_counters()
    lp.trace_closest(sc, rays)
    dbg = lp.debug_stats()
    row["max_leaf_undershoot"] = dbg.get("max_leaf_undershoot")
    row["max_leaf_undershoot_excess"] = dbg.get("max_leaf_undershoot_excess")
    lp.close()
    out[name] = row
    print(name, json.dumps(row), flush=True)


def main():
    out = {}
    rng = np.random.default_rng(7)
    angles = (0.3, 0.1, 0.03, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6)
    for asp in ((1.0, 2.0, 4.0), (10.0, 40.0, 110.0), (300.0, 1000.0), (1e4, 1e5, 1e6)):
        sc, tris, transforms = adv.needle_scene(pkg, rng, 20000, True, False, aspects=asp, degenerate=0.0)
        parts, labels = [], []
        for a in angles:
            r = adv.grazing_rays(abi, rng, adv.world_tris(tris, transforms[0]), 2, angles=(a,))
            parts.append(r); labels.append(np.full(len(r), a))
        r = adv.random_rays(abi, rng, 2000000, 5.0)
        parts.append(r); labels.append(np.full(len(r), -1.0))
        run(f"needles_aspect_{int(asp[0])}_{int(asp[-1])}", sc, np.concatenate(parts), out, np.concatenate(labels))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), "gpurun_out", "adversarial_probe.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
