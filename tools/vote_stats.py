#!/usr/bin/env python3
"""Measurement only: the wave vote of the per-ray kernels over a whole frame of the bench scene (counted kernels with
PTX_COUNT_FORWARD=1, i.e. the work the timed kernels do).  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTX_COUNT_FORWARD"] = "1"
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package()
    name = sys.argv[1] if len(sys.argv) > 1 else "bistro_like"
    bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    scene = pkg.Scene.procedural(name, 1, 1.0, 0 if name == "instanced_stress" else 1)
    w, h, spp = 1920, 1080, 16
    cam = scene.default_camera(w / h)
    loop = pkg.HipRenderLoop(0)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    loop.set_counting(True)
    loop.reset_counters()
    fb.render_passes(scene, cam, pkg.PathTracer(bounces), 0, spp)
    loop.sync()
    c = loop.counters()
    d = loop.debug_stats()
    loop.set_counting(False)
    n = max(1, c["rays_closest"])
    print(f"{name}: per closest-hit ray {c['node_visits'] / (n + c['rays_any']):.1f} node visits (all rays), {c['instance_entries'] / (n + c['rays_any']):.1f} instance entries, {c['triangle_tests'] / (n + c['rays_any']):.1f} triangle tests")
    so = d["stack_ops"]
    print(f"stack: {so[0]} pops, {so[1] / max(1, so[0]):.3f} of them from the HBM strips; {so[2]} child pushes, {so[3] / max(1, so[2]):.3f} to the strips")
    names = ("NODE", "TRI", "ENTER", "EXIT")
    for label, v, rays in (("closest", d["vote_closest"], c["rays_closest"]), ("any-hit", d["vote_any"], c["rays_any"])):
        print(f"{label}: {rays} rays, {v[0]} wave iterations = {v[0] * 64 / max(1, rays):.1f} lane-iterations per ray")
        for k in range(4):
            print(f"   {names[k]:5s} runs in {v[1 + k] / max(1, v[0]):.3f} of iterations with {v[5 + k] / max(1, v[1 + k]):5.1f} lanes   ({v[5 + k] / max(1, rays):.1f} per ray)")


if __name__ == "__main__":
    main()
