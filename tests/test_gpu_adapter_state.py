"""State the adapter and the library keep BETWEEN calls (the reference keeps none: marching_cubes.cu:229-263 counts and
allocates per call):

  * output-size hints of the pybind adapter (csrc/bindings.cpp): the largest counts of the last four calls on a shape, so a
    sparse frame does not make the next dense frame stream the field twice (the hint was the LAST call's);
  * the region totals of the last two calls on a shape (since round 6): when they hardly moved, the next call stores its
    vertices where they stay (the predicted region layout, p3d_mc_slab.region_first_rows) instead of through a scratch tensor;
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


@pytest.mark.parametrize("shape", [(72, 1024, 1024), (160, 1024, 512)])
def test_a_surface_inside_one_band_of_y_tiles_spreads_over_all_cursor_groups(gpu, built, shape):
    """A wall at a constant y in a 1024-wide grid (ncol = 172 / 86 tile columns: even, not a multiple of 8) lives in ONE of
    the 8 y bands the XCDs own.  Until round 5 such a band reached 2 (4) of the 8 cursor groups, its 8 (16) regions filled
    four times (twice) as fast as their share of the scratch allows, overflowed on every call, the field was streamed twice
    and the adapter's slack doubled for good.  The cursor group now rotates with the slab: one pass per call,
    from the first call on, and the slack stays where it was."""
    from primitive3d_amd import capi
    from tests.test_gpu_configs import torch_counts
    rx, ry, rz = shape
    y = torch.arange(ry, device=gpu, dtype=torch.float32)
    # a wavy wall around y = 500: |y - 500.4| stays inside one band of 126 / 252 rows; x and z wiggle it by < 1 voxel
    g = (y[None, :, None] - 500.4
         + 0.3 * torch.sin(torch.arange(rx, device=gpu)[:, None, None] * 0.7)
         + 0.3 * torch.cos(torch.arange(rz, device=gpu)[None, None, :] * 0.3)).contiguous()
    want = torch_counts(g, 0.0)
    assert want[0] > 32 * 8192 // 4   # enough vertices that 8 regions of the first call's scratch would not hold them
    upper = [float(s_) for s_ in shape]
    for i in range(3):
        got, how = _call(built, g, upper)
        assert got == want and how[0] == 1, (i, got, want, how)   # ONE streaming pass, every call
    # and spread it is: no region of the (hinted) scratch overflowed on the way
    ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device=gpu)
    v = torch.empty((want[0] + 64, 3), device=gpu)
    f = torch.empty((want[1] + 64, 3), dtype=torch.int32, device=gpu)
    capi.extract_fused_raw(g, 0.0, [0.0] * 3, upper, ws, v, f)
    nv, nf, flags = capi.read_counts(ws, with_flags=True)
    hdr = ws[:8192].view(torch.int64).cpu()
    regions = hdr[32:32 + 32 * 16:16]   # the 32 region totals the finishing block leaves in the header
    assert (nv, nf) == want and flags == 0 and int(regions.sum()) == nv
    assert int((regions > 0).sum()) == 32 and int(regions.max()) < 3 * nv // 32, regions.tolist()


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


def _layout_passes():
    from primitive3d_amd import capi
    return capi.debug_counters()["layout_passes"]


def _same_mesh(a, b):
    from bench import soup_hashes
    return all(torch.equal(x, y) for x, y in zip(soup_hashes(*a), soup_hashes(*b)))


def test_a_field_that_stands_still_is_laid_out_from_the_third_call_on(gpu, built):
    """Two calls on a shape give the adapter two sets of region totals; when they agree the third call lays its 32 regions
    out inside the vertex tensor itself (one layout pass per call, no scratch) -- and returns the mesh the first call did,
    vertex for vertex and triangle for triangle (as sorted soups: the row ORDER depends on which wave came first, in
    every mode)."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shape = (131, 120, 200)   # (a shape no other test uses)
    g = perlin_grid(shape, period=32, seed=11, device=gpu)
    upper = [float(s_) for s_ in shape]
    want = capi.extract(g, 0.0, [0.0] * 3, upper)   # the two-pass route into exactly sized tensors
    seen = []
    for i in range(6):
        l0, (p0, _) = _layout_passes(), _passes()
        v, f = built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, upper)
        seen.append((_layout_passes() - l0, _passes()[0] - p0))
        assert (v.shape[0], f.shape[0]) == (want[0].shape[0], want[1].shape[0]), i
        assert int(f.max()) < v.shape[0] and _same_mesh((v, f), want[:2]), i
    # (the first call on a shape may need a second pass: this field is denser than the first guess of a vertex per 16 voxels)
    assert seen[0] in ((0, 1), (0, 2)) and seen[1:] == [(0, 1)] + [(1, 1)] * 4, seen


def test_a_slowly_changing_field_stays_laid_out(gpu, built):
    """The iso level creeps up by 0.003 per call (a field under optimisation, extracted every few steps): every region's
    total moves by a per cent or so, the regions that grew spill into the eight areas behind the last region, the rows that
    end up beyond V are moved down -- one pass per call, a layout pass from the third call on, and every mesh is the
    two-pass route's."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shape = (96, 200, 136)
    g = perlin_grid(shape, period=32, seed=12, device=gpu)
    upper = [float(s_) for s_ in shape]
    laid = 0
    for i in range(10):
        t = -0.015 + 0.003 * i
        want = capi.extract(g, t, [0.0] * 3, upper)
        l0, (p0, _) = _layout_passes(), _passes()
        v, f = built.libPrim3D.marching_cubes(g, t, [0.0] * 3, upper)
        laid += _layout_passes() - l0
        assert _passes()[0] - p0 == 1, i
        assert (v.shape[0], f.shape[0]) == (want[0].shape[0], want[1].shape[0]) and _same_mesh((v, f), want[:2]), i
    assert laid == 8, laid


def test_a_jumping_field_goes_through_the_scratch(gpu, built):
    """Two fields of quite different density in turn: a layout made from one would be off by a third for the other (every
    region overflowing into the spill areas, every wave-plane asking two cursors), so the adapter does not try -- and when
    the field then stands still, it does again."""
    from primitive3d_amd.fields import perlin_grid
    from tests.test_gpu_configs import torch_counts
    shape = (90, 136, 200)
    a = perlin_grid(shape, period=32, seed=13, device=gpu)
    b = perlin_grid(shape, period=16, seed=14, device=gpu)
    wa, wb = torch_counts(a, 0.0), torch_counts(b, 0.0)
    assert wb[0] > 1.3 * wa[0]
    upper = [float(s_) for s_ in shape]
    l0 = _layout_passes()
    for i in range(8):
        v, f = built.libPrim3D.marching_cubes(b if i & 1 else a, 0.0, [0.0] * 3, upper)
        assert (v.shape[0], f.shape[0]) == (wb if i & 1 else wa), i
    assert _layout_passes() == l0
    for i in range(4):
        v, f = built.libPrim3D.marching_cubes(b, 0.0, [0.0] * 3, upper)
        assert (v.shape[0], f.shape[0]) == wb, i
    assert _layout_passes() == l0 + 3   # (all but the first of these: the jumping run ended on b, so that one still saw a, b)


def test_a_field_that_jumps_after_standing_still_overflows_the_spill_areas_once(gpu, built):
    """a, a, a, then b -- a third denser -- from the fourth call on.  The fourth call is laid out from a's totals (a stood
    still): b's surplus does not fit into the spill areas (V/8 + 4096 rows), the library flags it (bit 2), the adapter emits
    the mesh again into exactly sized tensors (a second streaming pass) and sits out two calls on the scratch route before
    it lays b out from b's totals.  Every call returns the right mesh."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shape = (100, 136, 192)
    a = perlin_grid(shape, period=32, seed=15, device=gpu)
    b = perlin_grid(shape, period=16, seed=16, device=gpu)
    upper = [float(s_) for s_ in shape]
    wa, wb = capi.extract(a, 0.0, [0.0] * 3, upper), capi.extract(b, 0.0, [0.0] * 3, upper)
    assert wb[0].shape[0] > 1.3 * wa[0].shape[0]
    seen = []
    for i, (g, want) in enumerate([(a, wa)] * 3 + [(b, wb)] * 5):
        l0, s0 = _layout_passes(), capi.debug_counters()["streaming_launches"]
        v, f = built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, upper)
        seen.append((_layout_passes() - l0, capi.debug_counters()["streaming_launches"] - s0))
        assert (v.shape[0], f.shape[0]) == (want[0].shape[0], want[1].shape[0]) and _same_mesh((v, f), want[:2]), i
        if i == 3:   # (the re-emission: exactly sized, freshly allocated tensors)
            assert v.untyped_storage().nbytes() == v.shape[0] * 12 and f.untyped_storage().nbytes() == f.shape[0] * 12
    # (laid out?, launches of the streaming kernel)
    assert seen[0] in ((0, 1), (0, 2)) and seen[1:3] == [(0, 1), (1, 1)], seen
    assert seen[3] == (1, 2), seen                      # laid out from a's totals, overflowed, streamed again by p3d_mc_emit
    assert seen[4:6] == [(0, 1), (0, 1)], seen          # two calls through the scratch
    assert seen[6:] == [(1, 1), (1, 1)], seen           # b stands still: laid out from b's totals


def test_threads_sharing_a_shape_share_its_hints(gpu, built):
    """Three host threads extract three different fields of ONE shape through the adapter (its hints are per shape, process-wide):
    whatever the interleaving -- a thread's call may be laid out from another thread's field, spill, overflow and be emitted
    again -- every call returns its own field's mesh."""
    import threading
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shape = (88, 136, 200)
    upper = [float(s_) for s_ in shape]
    fields = [perlin_grid(shape, period=32, seed=21, device=gpu), perlin_grid(shape, period=32, seed=22, device=gpu),
              perlin_grid(shape, period=16, seed=23, device=gpu)]
    wants = [capi.extract(g, 0.0, [0.0] * 3, upper) for g in fields]
    torch.cuda.synchronize()
    errors = []
    l0 = _layout_passes()

    def work(k):
        try:
            with torch.cuda.device(gpu):
                for i in range(80):
                    v, f = built.libPrim3D.marching_cubes(fields[k], 0.0, [0.0] * 3, upper)
                    assert (v.shape[0], f.shape[0]) == (wants[k][0].shape[0], wants[k][1].shape[0]), (k, i)
                    if i % 16 == 0:
                        assert _same_mesh((v, f), wants[k][:2]), (k, i)
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert _layout_passes() > l0   # (runs of one thread's calls in a row do get laid out)


def test_the_1024_cubed_volume_laid_out_equals_its_scratch_route(gpu, built):
    """BASELINE.json's C4 volume on one GPU, the way bench.py's `c4_1gpu` calls it: the third call stores 42 M vertices where they
    stay (regions of 1.3 M rows each, ids far beyond the 2^26 a region can number on the scratch route).  Same counts as an
    independent torch count, and the same mesh as the first call's -- vertex for vertex, triangle for triangle (sorted soups)."""
    from bench import soup_hashes
    from primitive3d_amd.fields import perlin_grid
    from tests.test_gpu_configs import torch_counts
    shape = (1024, 1024, 1024)
    g = perlin_grid(shape, period=64, seed=0, device=gpu)
    want = torch_counts(g, 0.0)
    upper = [float(s_) for s_ in shape]
    l0 = _layout_passes()
    v, f = built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, upper)
    assert (v.shape[0], f.shape[0]) == want and _layout_passes() == l0
    first = soup_hashes(v, f)
    del v, f
    built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, upper)
    v, f = built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, upper)
    assert (v.shape[0], f.shape[0]) == want and _layout_passes() == l0 + 1
    assert int(f.max()) == v.shape[0] - 1 and int(f.min()) == 0
    third = soup_hashes(v, f)
    assert all(torch.equal(a, b) for a, b in zip(first, third))


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
    ring_bytes0 = capi.debug_counters()["ring_bytes"]
    free0 = torch.cuda.mem_get_info(gpu)[0]
    for _ in range(1000):
        job()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(gpu)[0]
    assert rings() == rings0   # every stream's ring went with p3d_mc_release_stream
    # the rings' device memory itself, by the library's own account (out[6] of p3d_mc_debug_counters): freed, not only
    # dropped from the map -- a ring is 64 KiB, so the free-memory figure below could not tell
    assert capi.debug_counters()["ring_bytes"] == ring_bytes0, (capi.debug_counters(), ring_bytes0)
    # (memory: logged, with a generous one-sided bound -- other processes and earlier tests move the figure too)
    assert free0 - free1 <= 64 << 20, (free0, free1)
    # 100 streams alive at once, none released: their rings are there ...
    live = [create() for _ in range(100)]
    for h in live:
        run_on(h)
    torch.cuda.synchronize()
    assert rings() == rings0 + 100 and capi.debug_counters()["ring_bytes"] == ring_bytes0 + 100 * 16 * 4096
    # ... until they are released, one by one (the first half) or all at once by p3d_mc_shutdown (the rest)
    for h in live[:50]:
        capi.release_stream(h.value)
    assert rings() == rings0 + 50
    capi.shutdown()
    assert rings() == 0 and capi.debug_counters()["ring_bytes"] == 0
    for h in live:
        assert hip.hipStreamDestroy(h) == 0
    # and the library works again afterwards (its state is created on first use)
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [6, 8, 70], ws, v, f, scratch=scratch)
    assert capi.read_counts(ws) == want
