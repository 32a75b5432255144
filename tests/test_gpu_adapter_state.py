"""State the adapter and the library keep BETWEEN calls (the reference keeps none: marching_cubes.cu:229-263 counts and
allocates per call):

  * output-size hints of the pybind adapter (csrc/bindings.cpp): the largest counts of the last four calls on a shape, so a
    sparse frame does not make the next dense frame stream the field twice (the hint was the LAST call's);
  * the per-stream cursor ring of libp3dmc.so: p3d_mc_release_stream / p3d_mc_shutdown give it back.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _passes():
    from primitive3d_amd import capi
    c = capi.debug_counters()
    return c["streaming_passes"], c["emissions_without_a_pass"]


def _call(built, g, upper):
    p0, e0 = _passes()
    v, f = built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, upper)
    p1, e1 = _passes()
    return (v.shape[0], f.shape[0]), (p1 - p0, e1 - e0)


def test_alternating_sparse_and_dense_frames_stream_once(gpu, built):
    """sparse / dense / sparse / dense ... on one shape: EVERY call is one streaming pass.  The first dense frame has nothing
    but an empty frame to go by for its output buffers -- but the vertex scratch is sized for at least a vertex per 16 voxels,
    so only the buffers are too small: faces and compaction run a second time (p3d_mc_slab.part = 6), the field is not streamed
    twice.  From the third call on the buffers fit as well."""
    from primitive3d_amd.fields import perlin_grid
    from tests.test_gpu_configs import torch_counts
    shape = (250, 256, 256)   # (a shape no other test uses: its hints start empty)
    dense = perlin_grid(shape, period=64, seed=3, device=gpu)
    sparse = torch.ones(shape, device=gpu)   # all outside: V = F = 0
    want = torch_counts(dense, 0.0)
    assert 100000 < want[0] < shape[0] * shape[1] * shape[2] // 16
    upper = [float(s) for s in shape]
    per_call = []
    for i in range(8):
        got, how = _call(built, dense if i & 1 else sparse, upper)
        per_call.append(how)
        assert got == (want if i & 1 else (0, 0)), i
    assert per_call[1] == (1, 1), per_call          # one pass, emitted twice
    assert per_call[2:] == [(1, 0)] * 6, per_call   # one pass, emitted once


def test_dense_size_is_forgotten_after_four_sparse_frames(gpu, built):
    from primitive3d_amd.fields import perlin_grid
    shape = (130, 256, 320)
    dense = perlin_grid(shape, period=64, seed=4, device=gpu)
    sparse = torch.ones(shape, device=gpu)
    upper = [float(s) for s in shape]
    seq = [dense, dense, sparse, sparse, sparse, dense, sparse, sparse, sparse, sparse, dense]
    per_call = [_call(built, g, upper)[1] for g in seq]
    assert per_call[1] == (1, 0) and per_call[5] == (1, 0), per_call   # three sparse frames in between: still remembered
    assert per_call[10] == (1, 1), per_call                              # four: forgotten (buffers shrink again) -> emitted twice


def test_a_field_denser_than_the_scratch_is_streamed_twice(gpu, built):
    """White noise has a vertex on nearly every other edge: far beyond a vertex per 16 voxels.  Its first call overflows the
    scratch regions as well as the buffers, and only a second pass over the field (into exactly sized buffers) can help."""
    from tests.test_gpu_configs import torch_counts
    shape = (60, 64, 130)
    g = torch.from_numpy(np.random.default_rng(5).standard_normal(shape).astype(np.float32)).to(gpu)
    want = torch_counts(g, 0.0)
    assert want[0] > shape[0] * shape[1] * shape[2] // 4
    upper = [float(s) for s in shape]
    got, how = _call(built, g, upper)
    assert got == want and how == (2, 0), (got, how)
    got, how = _call(built, g, upper)
    assert got == want and how == (1, 0), (got, how)


def _hip():
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    return lib


def test_release_stream_keeps_device_memory_flat(gpu, built):
    """A C caller that creates a stream per job: 1000 streams, one extraction on each, p3d_mc_release_stream before the
    stream is destroyed -> the library keeps no ring for it any more (p3d_mc_debug_counters: `stream_rings`; a ring is
    64 KiB of device memory since round 5, 2 MiB before).  Without the release the rings stay (that is what the entry
    point is for)."""
    from primitive3d_amd import capi
    hip = _hip()
    g = torch.from_numpy(np.random.default_rng(0).standard_normal((6, 8, 70)).astype(np.float32)).to(gpu)
    ws = torch.empty(capi.workspace_bytes(*g.shape), dtype=torch.uint8, device=gpu)
    v = torch.empty((4096, 3), device=gpu)
    f = torch.empty((8192, 3), dtype=torch.int32, device=gpu)
    scratch = torch.empty((capi.scratch_rows_for(4096), 3), device=gpu)
    want = None

    def run_on(h):
        nonlocal want
        with torch.cuda.stream(torch.cuda.ExternalStream(h.value, device=gpu)):
            capi.extract_fused_raw(g, 0.0, [0, 0, 0], [6, 8, 70], ws, v, f, scratch=scratch)
            got = capi.read_counts(ws)
        want = want or got
        assert got == want and got[0] > 0

    def create():
        h = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(h)) == 0
        return h

    def job():
        h = create()
        run_on(h)
        capi.release_stream(h.value)
        assert hip.hipStreamDestroy(h) == 0

    for _ in range(20):
        job()   # (warm-up: allocator pools, the mailbox)
    torch.cuda.synchronize()
    rings = lambda: capi.debug_counters()["stream_rings"]
    rings0 = rings()
    free0 = torch.cuda.mem_get_info(gpu)[0]
    for _ in range(1000):
        job()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(gpu)[0]
    assert rings() == rings0   # every stream's ring went with p3d_mc_release_stream
    # (memory: logged, with a generous one-sided bound -- other processes and earlier tests move the figure too)
    assert free0 - free1 <= 64 << 20, (free0, free1)
    # 100 streams alive at once, none released: their rings are there ...
    live = [create() for _ in range(100)]
    for h in live:
        run_on(h)
    torch.cuda.synchronize()
    assert rings() == rings0 + 100
    # ... until they are released, one by one (the first half) or all at once by p3d_mc_shutdown (the rest)
    for h in live[:50]:
        capi.release_stream(h.value)
    assert rings() == rings0 + 50
    capi.shutdown()
    assert rings() == 0
    for h in live:
        assert hip.hipStreamDestroy(h) == 0
    # and the library works again afterwards (its state is created on first use)
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [6, 8, 70], ws, v, f, scratch=scratch)
    assert capi.read_counts(ws) == want
