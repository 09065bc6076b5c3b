"""The entry point that allocates nothing (`ptx_trace_batch_noalloc`, include/ptx_core.h) against `ptx_trace_batch`.

Collected last on purpose (`test_zz_`): the round's end runs `pytest -x`, and a failure here must not hide other tests.  Inputs and outputs are device buffers the
TEST allocates (hipMalloc through ctypes); the entry point itself makes no hipMalloc / hipFree — its scratch (queue cursors, error
word, stack spill strips) is a file-scope device array resolved once per device with hipGetSymbolAddress.
Two calls in a row, the second with a smaller input: the array keeps the first call's contents, so the second call
proves that what must start at zero is cleared.  Bar: bit-exact (integer ids, f32 bits, u8 flags).
"""
import numpy as np
import pytest

from hip_ctypes import Hip

pytestmark = pytest.mark.gpu


def test_noalloc_trace_batch_matches_trace_batch(pkg):
    abi = pkg._abi
    loop = pkg.HipRenderLoop(0)
    sc = pkg.Scene.procedural("sponza_like", 1, 0.3, 1)      # deep enough that rays spill into the stack strips
    w, h = 320, 180
    cam = sc.default_camera(w / h)
    prim = loop.primary_rays(sc, cam, w, h, 1)               # binds (uploads) the scene: the only allocation, before the calls under test
    hits = loop.trace_closest(sc, prim)
    ok = hits["inst_idx"] != abi.PTX_MISS
    rng = np.random.default_rng(5)
    o = prim["origin"][ok] + prim["direction"][ok] * hits["t"][ok, None]
    d = rng.normal(size=o.shape).astype(np.float32)
    bounce = np.zeros(len(o), dtype=abi.ray_dtype())
    bounce["origin"], bounce["direction"], bounce["t_max"] = o + d * np.float32(1e-3), d, np.inf
    bounce["t_max"][::3] = rng.uniform(0.05, 3.0, size=len(bounce["t_max"][::3])).astype(np.float32)
    big = np.concatenate([prim, bounce])
    small = np.ascontiguousarray(big[1000:1000 + 64 * 37 + 5])    # second call: smaller, ragged last wave

    hip = Hip()
    try:
        for rays in (big, small):
            want_hits = loop.trace_closest(sc, rays)
            want_any = loop.trace_any(sc, rays)
            d_rays = hip.upload(rays)
            d_hits = hip.malloc(len(rays) * abi.hit_dtype().itemsize)
            d_any = hip.malloc(len(rays))
            loop.sync()
            before = hip.free_bytes()
            loop.trace_batch_noalloc(sc, d_rays, len(rays), abi.PTX_TRACE_CLOSEST, d_out_hits_ptr=d_hits)
            loop.trace_batch_noalloc(sc, d_rays, len(rays), abi.PTX_TRACE_ANY, d_out_any_ptr=d_any)
            assert hip.free_bytes() >= before, "device memory was allocated during the no-alloc calls"
            got_hits = hip.download(d_hits, abi.hit_dtype(), len(rays))
            got_any = hip.download(d_any, np.uint8, len(rays))
            assert np.array_equal(got_hits.view(np.uint8), want_hits.view(np.uint8)), "hit records differ from ptx_trace_batch"
            assert np.array_equal(got_any, want_any), "any-hit flags differ from ptx_trace_batch"
            assert (want_hits["inst_idx"] != abi.PTX_MISS).mean() > 0.3 and 0.05 < want_any.mean() < 1.0

        # sizes beyond the stated maximum are refused with a clear error, nothing is launched
        with pytest.raises(pkg.RenderCoreError, match="PTX_NOALLOC_MAX_RAYS"):
            loop.trace_batch_noalloc(sc, d_rays, (1 << 24) + 1, abi.PTX_TRACE_CLOSEST, d_out_hits_ptr=d_hits)
        # n == 0 is a no-op
        loop.trace_batch_noalloc(sc, d_rays, 0, abi.PTX_TRACE_CLOSEST, d_out_hits_ptr=d_hits)
    finally:
        hip.release()
