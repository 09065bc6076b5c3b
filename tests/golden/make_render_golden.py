"""Regenerates tests/golden/render_vectors.json from the CPU oracle's deterministic-math build: per test scene the
sha256 of the primary rays, their closest-hit records and any-hit flags, and of one render_pass image (f32 bits).
The reference ships no fixtures and cannot be built here (no Rust toolchain), so these are regression pins of the
restatement, not outputs of the reference ("parity unpinned", oracle/README).  Run from the repo root:
    python tests/golden/make_render_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import oracle_py  # noqa: E402

CASES = [
    # name, flags, detail, w, h, bounces
    ("cornell", 0, 1.0, 96, 54, 8),
    ("cornell", 1, 1.0, 64, 36, 16),
    ("helmet_like", 0, 0.05, 96, 54, 8),
    ("sponza_like", 1, 0.05, 96, 54, 16),
    ("instanced_stress", 0, 0.05, 96, 54, 4),
    ("glass_test", 1, 0.3, 64, 36, 16),
    ("sss_test", 1, 0.3, 64, 36, 16),
    ("bistro_like", 3, 0.02, 96, 54, 16),
]
PRIMARY_SAMPLE, IMAGE_SAMPLE = 3, 5


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_key(name, flags, detail, w, h, bounces):
    return f"{name}/flags{flags}/detail{detail}/{w}x{h}/b{bounces}"


def vectors(trace_closest, trace_any, render_pass, rays):
    hits = trace_closest(rays)
    return {
        "primary_rays": sha(rays.view(np.uint8)),
        "closest_hits": sha(hits.view(np.uint8)),
        "any_hits": sha(np.asarray(trace_any(rays), dtype=np.uint8)),
        "image_f32": sha(np.asarray(render_pass(), dtype=np.float32)),
    }


def main():
    pkg = ge.load_package()
    oracle = oracle_py.Oracle("det")
    out = {}
    for name, flags, detail, w, h, bounces in CASES:
        sc = pkg.Scene.procedural(name, 1, detail, flags)
        cam = sc.default_camera(w / h)
        osc = oracle.scene(sc)
        rays = osc.primary_rays(cam, w, h, PRIMARY_SAMPLE)
        out[case_key(name, flags, detail, w, h, bounces)] = vectors(
            osc.trace_closest, osc.trace_any, lambda: osc.render_pass(cam, w, h, IMAGE_SAMPLE, bounces), rays)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "render_vectors.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
