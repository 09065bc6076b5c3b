#!/usr/bin/env python3
"""`--benchmark` of the reference (src/main.rs:59-153) over the HIP render core, for a real .glb/.gltf file or one of
the procedural stand-ins:   python tools/pt_benchmark.py --benchmark [model.glb | scene_name]
Same defaults (640x360, 16 samples, PathTracer<8>, solid sky (0.5,0.6,0.8), no sun, the file's first perspective camera
or eye (0,1,3) -> (0,0.5,0) fov 60) and the same report lines.  lib/pt_bench is the C++ twin for procedural scenes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def run_benchmark(args):
    pkg = ge.load_package()
    target = next((a for a in args[1:] if not a.startswith("-")), "assets/models/bistro.glb")
    print(f"Benchmark: {target}")
    print("Loading model...")
    t0 = time.perf_counter()
    width, height = 640, 360
    aspect = width / height
    if target.lower().endswith((".glb", ".gltf")):
        g = pkg.gltf.load_gltf(target)
        load_time = time.perf_counter() - t0
        print(f"  Loaded in {load_time:.3f}s: {len(g.instances)} instances, {len(g.materials)} materials, {len(g.textures)} textures")
        print("Building acceleration structure...")
        t1 = time.perf_counter()
        scene = pkg.gltf.build_scene(g)
        camera = pkg.gltf.camera_for(g, aspect)
    else:
        scene = pkg.Scene.procedural(target, 1, 1.0, 0)
        load_time = time.perf_counter() - t0
        d = scene.desc.contents
        print(f"  Loaded in {load_time:.3f}s: {d.n_instances} instances, {d.n_materials} materials, {d.n_textures} textures")
        print("Building acceleration structure...")
        t1 = time.perf_counter()
        camera = scene.default_camera(aspect)
    loop = pkg.HipRenderLoop(0)
    loop._bind(scene, camera)
    build_time = time.perf_counter() - t1
    print(f"  Built in {build_time:.3f}s")
    renderer = pkg.PathTracer(8)
    fb = pkg.DeviceFramebuffer(loop, width, height, adaptive=False)
    num_samples = 16
    print(f"Rendering {num_samples} samples at {width}x{height}...")
    fb.reset()
    loop.sync()
    start = time.perf_counter()
    for sample in range(0, num_samples, 4):
        fb.render_passes(scene, camera, renderer, sample, 4)
        loop.sync()
        print(f"  Sample {sample + 4}/{num_samples}: {(sample + 4) / (time.perf_counter() - start):.2f} spp/sec")
    mean = fb.to_color_buffer()
    render_time = time.perf_counter() - start
    total = width * height * num_samples
    print()
    print("Results:")
    print(f"  Build time:   {build_time:.3f}s")
    print(f"  Render time:  {render_time:.3f}s")
    print(f"  Primary rays: {total / 1e6:.2f}M")
    print(f"  Performance:  {total / render_time / 1e6:.2f} Mprimary/s")
    print(f"  Samples/sec:  {num_samples / render_time:.2f}")
    return mean


if __name__ == "__main__":
    if "--benchmark" not in sys.argv:
        raise SystemExit(__doc__)
    run_benchmark(sys.argv)
