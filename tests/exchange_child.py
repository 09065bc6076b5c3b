"""Child process of tests/test_gpu_multiprocess.py (not collected by pytest): one rank of a real RCCL process group
(backend "nccl", world_size 1 — RCCL refuses two ranks on one GPU) drives `TileExchange.combine` twice in a row with no
host synchronisation in between, then dumps what the all-gather delivered and the framebuffer, for the parent to compare
with the numpy mirror of the export kernel.
    python tests/exchange_child.py OUT.npz PORT"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, port = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = ge.load_package()
    w, h, spp = 200, 120, 4
    scene = pkg.Scene.procedural("cornell", 1, 1.0, 1)
    cam = scene.default_camera(w / h)
    loop = pkg.HipRenderLoop(0)
    fb = pkg.DeviceFramebuffer(loop, w, h, adaptive=True)
    ex = pkg.render.TileExchange(dist, loop, w, h, 0, 1)
    renderer = pkg.PathTracer(6)
    # two frames back to back: the second export may only overwrite `send` after the first collective has read it
    for first in (0, spp):
        fb.reset()
        fb.render_passes(scene, cam, renderer, first, spp)
        ex.combine(fb)
    loop.sync()
    torch.cuda.synchronize()
    s, q, c = fb.read()
    np.savez(out, recv=ex.recv.cpu().numpy(), s=s, q=q, c=c, w=w, h=h)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
