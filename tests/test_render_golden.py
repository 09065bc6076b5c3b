"""Committed golden vectors (tests/golden/render_vectors.json, made by tests/golden/make_render_golden.py from the
oracle's deterministic-math build): the oracle must keep reproducing them on the CPU, and the HIP render core must
produce the same bytes through the C ABI on the GPU.  They pin the restatement and the kernels against silent drift;
they are not outputs of the reference (it ships no fixtures and cannot be built here: "parity unpinned")."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_render_golden as mg  # noqa: E402

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "render_vectors.json")))


def test_golden_covers_every_case():
    assert sorted(GOLDEN) == sorted(mg.case_key(*c) for c in mg.CASES)
    for v in GOLDEN.values():
        assert sorted(v) == ["any_hits", "closest_hits", "image_f32", "primary_rays"]


@pytest.mark.parametrize("case", mg.CASES, ids=lambda c: mg.case_key(*c))
def test_oracle_reproduces_golden(pkg, oracle_det, case):
    name, flags, detail, w, h, bounces = case
    sc = pkg.Scene.procedural(name, 1, detail, flags)
    cam = sc.default_camera(w / h)
    osc = oracle_det.scene(sc)
    rays = osc.primary_rays(cam, w, h, mg.PRIMARY_SAMPLE)
    got = mg.vectors(osc.trace_closest, osc.trace_any, lambda: osc.render_pass(cam, w, h, mg.IMAGE_SAMPLE, bounces), rays)
    assert got == GOLDEN[mg.case_key(*case)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.CASES, ids=lambda c: mg.case_key(*c))
def test_gpu_reproduces_golden(pkg, loop, case):
    name, flags, detail, w, h, bounces = case
    sc = pkg.Scene.procedural(name, 1, detail, flags)
    cam = sc.default_camera(w / h)
    rays = loop.primary_rays(sc, cam, w, h, mg.PRIMARY_SAMPLE)
    got = mg.vectors(lambda r: loop.trace_closest(sc, r), lambda r: loop.trace_any(sc, r),
                     lambda: loop.render_pass(sc, cam, pkg.PathTracer(bounces), w, h, mg.IMAGE_SAMPLE), rays)
    assert got == GOLDEN[mg.case_key(*case)]
