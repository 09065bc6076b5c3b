"""The N > 1 path on CPU: two gloo ranks, each renders only the 8x8 tiles it owns (scene replicated), then the ranks
exchange their owned-tile records with ONE all-gather — the exchange bench.py runs over RCCL (`TileExchange`), here
with the numpy mirrors of `ptx_fb_export_tiles_dev` / `ptx_fb_import_tiles_dev` (`pack_tiles` / `unpack_tiles`; a GPU
test checks the kernels against these mirrors).  The pixel work is done by the oracle here (no GPU in this tier); what
is under test is the partition + exchange logic: the combined buffers must be bit-identical to a single-rank frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, bounces, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle_py import Oracle
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    scene = pkg.Scene.procedural("cornell", 1, 1.0, 0)
    cam = scene.default_camera(w / h)
    oracle = Oracle("det")
    oracle.set_threads(2)
    osc = oracle.scene(scene)
    owned = pkg.render.partition_pixels(w, h, rank, world)
    s = np.zeros((h * w, 3), dtype=np.float32)
    q = np.zeros((h * w, 3), dtype=np.float32)
    c = np.zeros(h * w, dtype=np.int32)
    for k in range(spp):
        work = np.stack([owned % w, owned // w, np.full(len(owned), k, dtype=np.uint32)], axis=1)
        px = osc.render_pixels(cam, w, h, work, bounces)
        s[owned] += px
        q[owned] += px * px
        c[owned] += 1
    # one all-gather of equal-sized blocks of packed owned-tile records, then scatter the other ranks' blocks
    mine = pkg.render.pack_tiles(w, h, rank, world, s, q, c.astype(np.uint32))
    assert mine.nbytes == pkg.render.tiles_bytes(w, h, world)
    blocks = [torch.empty(mine.nbytes, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(blocks, torch.from_numpy(mine.view(np.uint8).reshape(-1)))
    cu = c.astype(np.uint32)
    for r in range(world):
        if r != rank:
            pkg.render.unpack_tiles(w, h, r, world, blocks[r].numpy().view(np.uint32), s, q, cu)
    # the all-reduce form of the same exchange (every rank contributes zeros outside its tiles) must agree
    s2, q2, c2 = np.zeros_like(s), np.zeros_like(q), np.zeros_like(c)
    s2[owned], q2[owned], c2[owned] = s[owned], q[owned], c[owned]
    ts, tq, tc = torch.from_numpy(s2), torch.from_numpy(q2), torch.from_numpy(c2)
    pkg.render.combine_framebuffers(dist, (ts, tq, tc))
    assert np.array_equal(ts.numpy().view(np.uint32), s.view(np.uint32)) and np.array_equal(tc.numpy(), cu.astype(np.int32))
    if rank == 0:
        np.savez(os.path.join(out_dir, "combined.npz"), s=s, q=q, c=cu.astype(np.int32), n_owned=len(owned))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_disjoint_and_complete(pkg):
    for (w, h, world) in ((64, 36, 2), (100, 52, 3), (1920, 1080, 8), (7, 5, 2)):
        parts = [pkg.render.partition_pixels(w, h, r, world) for r in range(world)]
        allp = np.concatenate(parts)
        assert len(allp) == w * h and len(np.unique(allp)) == w * h
        if w * h > 10000:
            sizes = np.array([len(p) for p in parts])
            assert sizes.max() - sizes.min() <= 64 * 2 + 0.01 * sizes.mean()   # interleave balances the load
    # one wave (64 consecutive entries) is one tile
    p = pkg.render.partition_pixels(64, 32, 1, 2)[:64]
    assert (p % 64).max() - (p % 64).min() == 7 and (p // 64).max() - (p // 64).min() == 7


def test_two_rank_gloo_frame_equals_single_rank(pkg, oracle_det, tmp_path):
    import torch.multiprocessing as mp
    from oracle_py import OracleFramebuffer
    w, h, spp, bounces, world = 52, 27, 3, 8, 2      # not multiples of 8: ragged edge tiles, odd tile count
    port = _free_port()
    mp.spawn(_worker, args=(world, port, w, h, spp, bounces, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "combined.npz")
    scene = pkg.Scene.procedural("cornell", 1, 1.0, 0)
    cam = scene.default_camera(w / h)
    fb = OracleFramebuffer(oracle_det.scene(scene), w, h, True)
    fb.render_passes(cam, 0, spp, bounces)
    s, q, c = fb.read()
    assert 0 < int(got["n_owned"]) < w * h
    assert np.array_equal(got["c"].reshape(h, w), c.astype(np.int32))
    assert np.array_equal(got["s"].reshape(h, w, 3).view(np.uint32), s.view(np.uint32))
    assert np.array_equal(got["q"].reshape(h, w, 3).view(np.uint32), q.view(np.uint32))


def test_select_truncated_is_the_reference_cut(pkg):
    """`select_truncated` against the reference's own words (renderloop/parallel.rs:96-107): the noisy pixels in
    row-major order, a STABLE sort by error descending, truncate(max) — with many tied errors (the +inf of pixels below
    min_samples), whatever order the ranks' lists arrive in."""
    rng = np.random.default_rng(5)
    w, h = 37, 23
    err = rng.choice(np.array([0.0, 0.01, 0.02, 0.5, np.inf], dtype=np.float32), size=w * h)
    noisy = np.nonzero(err > 0.005)[0]
    # the reference: Vec of (pixel, err) in row-major order, sort_by(|a, b| b.err.partial_cmp(a.err)) (stable), truncate
    ref = sorted(noisy.tolist(), key=lambda p: -float(err[p]) if np.isfinite(err[p]) else -1e30)
    keys = err.view(np.uint32) + 1                         # k_noise_keys: f32 order of a non-negative error, +1
    for world in (1, 2, 3, 8):
        owner = pkg.render.tile_owner(noisy, w, world)
        chunks = [noisy[owner == r] for r in range(world)]
        for c in chunks:
            rng.shuffle(c)                                  # a rank's list is in no particular order
        allp = np.concatenate(chunks[::-1]).astype(np.uint32)
        for mx in (0, 1, 17, len(noisy) // 2, len(noisy), len(noisy) + 5):
            got = pkg.render.select_truncated(keys[allp], allp, mx)
            assert got.tolist() == ref[:mx]
    # tile_owner agrees with the partition the device uses
    for world in (2, 3):
        for r in range(world):
            mine = pkg.render.partition_pixels(w, h, r, world)
            assert (pkg.render.tile_owner(mine, w, world) == r).all()


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    n = [5, 0, 9][rank]                                       # ragged, one rank with nothing noisy
    keys = (np.arange(n, dtype=np.uint32) * 7 + rank + 0x7F800001 * (rank == 2)).astype(np.uint32)    # keys beyond 2^31 survive the trip
    pixels = np.arange(n, dtype=np.uint32) + 100 * rank
    k, x = pkg.render.gather_pairs_torch(dist)(keys, pixels)
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), k=k, x=x)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pairs_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want_k = np.concatenate([(np.arange(n, dtype=np.uint32) * 7 + r + 0x7F800001 * (r == 2)).astype(np.uint32) for r, n in enumerate([5, 0, 9])])
    want_x = np.concatenate([np.arange(n, dtype=np.uint32) + 100 * r for r, n in enumerate([5, 0, 9])])
    for r in range(world):
        got = np.load(tmp_path / f"g{r}.npz")
        assert got["k"].dtype == np.uint32 and np.array_equal(got["k"], want_k) and np.array_equal(got["x"], want_x)
