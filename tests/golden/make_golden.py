"""Regenerates tests/golden/host_tables.json: sha256 of tables/trees produced by THIS repo's host library.
They are regression pins (the reference ships no fixtures and cannot be run here); run from the repo root."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from test_host_build import random_boxes  # noqa: E402

pkg = ge.load_package()
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
b = pkg.host.Bvh8(random_boxes(1000, 7), 4)
cam = pkg.Camera.look_at((0.0, 1.0, 3.0), (0.0, 0.5, 0.0), 60.0, 640 / 360)
out = {
    "bluenoise_64_12345": sha(pkg.host.bluenoise_generate(64, 12345)),
    "bluenoise_64_67890": sha(pkg.host.bluenoise_generate(64, 67890)),
    "bvh8_random1000_nodes": sha(b.nodes()),
    "bvh8_random1000_prims": sha(b.primitive_indices()),
    "camera_benchmark_inv_vp": sha(np.array(list(cam.inv_vp_matrix), dtype=np.float32)),
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "host_tables.json"), "w"), indent=1)
print(out)
