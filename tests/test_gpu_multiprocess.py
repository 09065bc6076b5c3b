"""Multi-process GPU tests.  Every child is a FRESH process started before it touches the GPU; the parent only
launches them and compares the files they leave.

* bench.py's N > 1 road — one rank per process, tile partition, `TileExchange` (export kernel -> all-gather -> import
  kernels), the frame's own checks and the reference-order check — with two ranks on ONE card: gloo stands in for
  RCCL (which refuses two ranks on one GPU; `PTX_BENCH_BACKEND=gloo`, `PTX_BENCH_DEVICE=0`), everything else is the
  code the driver's 8-GPU run executes.  The combined sums must be the 1-rank frame's, bit for bit (same sha256).
* the RCCL flavour of the exchange with a real "nccl" process group of one rank: export -> all_gather_into_tensor ->
  ordering by stream events only, two frames in a row.
The data-parallel map being split is /root/reference/src/raytracer/renderloop/parallel.rs:185-192."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_ARGS = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--scene", "sponza_like", "--detail", "0.1",
              "--width", "328", "--height", "180", "--spp", "8", "--bounces", "8"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PTX_EXACT_ORDER", "PTX_NO_FORWARD", "PTX_PACKET_BATCH"):
        env.pop(k, None)
    env.update(extra)
    return env


def _last_json_line(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_two_rank_bench_frame_equals_the_one_rank_frame():
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *BENCH_ARGS], capture_output=True, text=True,
                         timeout=280, env=_env(), cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    a = _last_json_line(one.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *BENCH_ARGS],
                         capture_output=True, text=True, timeout=280, env=_env(PTX_BENCH_BACKEND="gloo", PTX_BENCH_DEVICE="0"), cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _last_json_line(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "strong"
    for r in (a, b):
        v = r["verify"]
        assert v["counts_eq_spp"] and v["finite"] and v["clamped"] and v["equals_reference_order"], v
    # ownership is disjoint and every pixel's samples are added in the same order: the same bits
    assert a["verify"]["sum_sha256_16"] == b["verify"]["sum_sha256_16"], (a["verify"], b["verify"])
    assert a["rays_per_frame"] == b["rays_per_frame"]
    assert "tiles8x8 mod 2" in b["config"]["parallelism"]


def test_rccl_exchange_orders_itself_by_stream_events(pkg, tmp_path):
    out = str(tmp_path / "exchange.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "exchange_child.py"), out, str(_free_port())],
                       capture_output=True, text=True, timeout=280, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    w, h = int(d["w"]), int(d["h"])
    want = pkg.render.pack_tiles(w, h, 0, 1, d["s"], d["q"], d["c"])
    assert want.nbytes == pkg.render.tiles_bytes(w, h, 1) == d["recv"].nbytes
    assert int(d["c"].min()) == int(d["c"].max()) == 4          # the second frame, not a mixture
    assert np.array_equal(d["recv"].view(np.uint32).reshape(-1, 7), want)


def test_two_contexts_rendering_alternate_frames_change_nothing(pkg):
    """`ptx_fb_render_passes_async` + `ptx_order_after_frame_tail` (bench.py --frames-in-flight 2): two contexts on one
    GPU render alternate frames, each starting once the other's frame is past its heavy iterations, with no host wait in
    between.  Every frame must be the frame one context renders alone, bit for bit (the map being pipelined is the
    `for sample` loop of /root/reference/src/main.rs:123-141, one frame after the other there)."""
    sc = pkg.Scene.procedural("sponza_like", 1, 0.1, 1)
    w, h, spp = 320, 184, 6
    cam = sc.default_camera(w / h)
    r = pkg.PathTracer(8)
    solo = pkg.HipRenderLoop(0)
    fb0 = pkg.DeviceFramebuffer(solo, w, h, adaptive=True)
    want = []
    for first in (0, 6, 12, 18):
        fb0.reset()
        fb0.render_passes(sc, cam, r, first, spp)
        want.append([a.copy() for a in fb0.read()])
    loops = [pkg.HipRenderLoop(0), pkg.HipRenderLoop(0)]
    fbs = [pkg.DeviceFramebuffer(lp, w, h, adaptive=True) for lp in loops]
    got = []
    for i, first in enumerate((0, 6, 12, 18)):
        k = i % 2
        if i >= 2:      # this context's previous frame: read it before it is overwritten (that read synchronises this context only)
            got.append([a.copy() for a in fbs[k].read()])
        loops[k].order_after_frame_tail(loops[1 - k])
        fbs[k].reset()
        fbs[k].render_passes(sc, cam, r, first, spp, wait=False)
    for k in (0, 1):
        loops[k].sync()
        got.append([a.copy() for a in fbs[k].read()])
    assert len(got) == 4
    for i in range(4):
        for a, b in zip(got[i], want[i]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    # a context cannot wait for itself, nor for a context that does not exist
    with pytest.raises(pkg.render.RenderCoreError):
        loops[0].order_after_frame_tail(loops[0])
