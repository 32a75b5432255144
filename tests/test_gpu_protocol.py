"""The order of an extraction's calls (p3d_mc_slab.part, include/p3d_mc.h) is CHECKED by the library: every successor the
table in the header does not have comes back as P3D_EINVAL with a message, synchronously, before anything is launched --
the reference's boundary is stateless (marching_cubes.h:14-15), a multi-call extraction must at least be hard to misuse.
And the per-stream cursor ring survives a call that fails between taking its block and its first launch."""
import os

import numpy as np
import pytest
import torch

from oracle import canonical_mesh, oracle_count, oracle_extract

pytestmark = pytest.mark.gpu

P3D_EINVAL = -1


class Rig:
    """One small grid with everything a part needs; call(part, ...) goes through the C ABI."""

    def __init__(self, gpu, shape=(40, 24, 70), seed=5):
        from primitive3d_amd import capi
        from primitive3d_amd.fields import perlin_grid
        self.capi = capi
        self.g = perlin_grid(shape, period=12, seed=seed).to(gpu)
        self.shape = shape
        self.ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device=gpu)
        self.nv, self.nf = oracle_count(self.g.cpu().numpy(), 0.0)
        self.scratch = torch.empty((capi.scratch_rows_for(self.nv), 3), device=gpu)
        self.other_scratch = torch.empty_like(self.scratch)
        self.v = torch.empty((self.nv, 3), device=gpu)
        self.other_v = torch.empty_like(self.v)
        self.f = torch.empty((self.nf, 3), dtype=torch.int32, device=gpu)
        self.lower, self.upper = [0.0, 0.0, 0.0], [float(n) for n in shape]

    def call(self, part, split=0, v=None, f=None, scratch="own", ws=None, grid=None):
        slab = self.capi.Slab()
        slab.part, slab.split_plane = part, split
        self.capi.extract_fused_raw(self.g if grid is None else grid, 0.0, self.lower, self.upper, self.ws if ws is None else ws,
                                    v, f, slab=slab, scratch=self.scratch if scratch == "own" else scratch)

    def count(self):
        self.capi.count(self.g, 0.0, self.ws)

    def emit(self, v=None, f=None):
        self.capi.emit(self.g, 0.0, self.lower, self.upper, self.ws, v, f)

    def refuse(self, fn, *needles):
        with pytest.raises(self.capi.P3DError) as ei:
            fn()
        assert ei.value.code == P3D_EINVAL, ei.value
        msg = str(ei.value)
        assert "illegal order of calls" in msg, msg
        for n in needles:
            assert n in msg, (n, msg)

    def fresh(self):
        """Forget what the workspace has seen: a starting call on another shape is not needed -- a fresh tensor is."""
        self.ws = torch.empty_like(self.ws)
        return self


def _mesh_ok(rig, v, f):
    from tests.ws_keys import vertex_keys_from_workspace
    torch.cuda.synchronize()
    keys = vertex_keys_from_workspace(rig.ws.cpu().numpy(), rig.shape, v.shape[0], rig.capi.debug_layout(*rig.shape))
    ref = canonical_mesh(*oracle_extract(rig.g.cpu().numpy(), 0.0))
    got = canonical_mesh(v.cpu().numpy(), f.cpu().numpy(), keys)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_every_illegal_successor_is_refused(gpu):
    r = Rig(gpu)
    split = 16
    # on a workspace that has seen nothing: every continuation
    for part, kw in [(2, dict(split=split)), (3, dict(split=split)), (4, {}), (5, {}), (6, dict(v=r.v, f=r.f))]:
        r.fresh()
        r.refuse(lambda: r.call(part, **kw), "none is in progress" if part != 6 else "")
    r.fresh()
    r.refuse(lambda: r.emit(r.v, r.f), "p3d_mc_emit needs")
    # after part 1: only 2 or 3 at the same split
    for part, kw, needle in [(4, {}, "part 4 needs part 3"), (5, {}, "part 5 needs part 4"), (6, dict(v=r.v, f=r.f), "part 6 needs"),
                             (2, dict(split=split + 1), "split_plane"), (3, dict(split=split + 2), "split_plane")]:
        r.fresh().call(1, split=split)
        r.refuse(lambda: r.call(part, **kw), needle, "part 1")
    r.fresh().call(1, split=split)
    r.refuse(lambda: r.emit(r.v, r.f), "finished counts")
    # after part 3: only 4
    for part, kw, needle in [(2, dict(split=split), "part 2 needs part 1"), (5, {}, "part 5 needs part 4"),
                             (6, dict(v=r.v, f=r.f), "part 6 needs"), (3, dict(split=split), "needs part 1")]:
        r.fresh().call(3)
        r.refuse(lambda: r.call(part, **kw), needle, "part 3")
    r.fresh().call(3)
    r.refuse(lambda: r.emit(r.v, r.f), "finished counts")
    # after part 4 WITHOUT a vertex buffer: part 5 may not write vertices (6 does); after part 4 WITH one: the same one
    r.fresh().call(3)
    r.call(4)
    r.refuse(lambda: r.call(5, v=r.v, f=r.f), "part 6 writes the vertices")
    r.refuse(lambda: r.call(2, split=split), "part 2 needs part 1")
    r.refuse(lambda: r.call(4), "part 4 needs part 3")
    r.fresh().call(3)
    r.call(4, v=r.v)
    r.refuse(lambda: r.call(5, v=r.other_v, f=r.f), "vertex buffer part 4 began to fill")
    r.refuse(lambda: r.call(5, v=None, f=r.f), "vertex buffer part 4 began to fill")
    # after a finished extraction: 6 (same scratch) and emit are fine, 2 / 4 / 5 are not
    r.fresh().call(0, v=r.v, f=r.f)
    for part, kw, needle in [(2, dict(split=split), "part 2 needs part 1"), (4, {}, "part 4 needs part 3"), (5, {}, "part 5 needs part 4")]:
        r.refuse(lambda: r.call(part, **kw), needle, "finished extraction")
    # p3d_mc_count starts anew: parts cannot continue it
    r.fresh().count()
    for part, kw in [(2, dict(split=split)), (4, {}), (5, {}), (6, dict(v=r.v, f=r.f))]:
        r.refuse(lambda: r.call(part, **kw), "p3d_mc_count")
    # the batched entry: nothing continues it
    grids = r.g[None].contiguous()
    wsb = torch.empty(r.capi.workspace_bytes_batched(1, *r.shape), dtype=torch.uint8, device=gpu)
    off = torch.zeros(4, dtype=torch.int64, device=gpu)
    r.capi.extract_fused_batched_raw(grids, 0.0, r.lower, r.upper, wsb, r.v, r.scratch, r.f, off)
    r.refuse(lambda: r.call(6, v=r.v, f=r.f, ws=wsb), "p3d_mc_extract_fused_batched")


def test_same_stream_shape_and_scratch_are_enforced(gpu):
    r = Rig(gpu)
    r.call(3)
    r.refuse(lambda: r.call(4, scratch=r.other_scratch), "scratch")
    s2 = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(s2):
        r.refuse(lambda: r.call(4), "same stream")
    smaller = r.g[:-1].contiguous()
    r.refuse(lambda: r.call(4, grid=smaller), "same grid shape")
    r.call(4)   # the refused calls changed nothing: the legal successor still is
    r.refuse(lambda: r.call(6, v=r.v, f=r.f, scratch=r.other_scratch), "scratch")
    r.call(6, v=r.v, f=r.f)
    _mesh_ok(r, r.v, r.f)
    s2.synchronize()
    r.capi.release_stream(s2.cuda_stream)


def test_the_legal_walks_give_the_oracles_mesh(gpu):
    r = Rig(gpu)
    split = 16
    # 1 -> 3 -> 4 -> 5
    r.call(1, split=split, v=r.v)
    r.call(3, split=split, v=r.v)
    r.call(4, v=r.v)
    assert r.capi.read_counts(r.ws) == (r.nv, r.nf)
    r.call(5, v=r.v, f=r.f)
    _mesh_ok(r, r.v, r.f)
    # 1 -> 2 (finished) -> 6 into other buffers -> 6 again
    r.fresh().call(1, split=split, v=r.v)
    r.call(2, split=split, v=r.v, f=r.f)
    assert r.capi.read_counts(r.ws) == (r.nv, r.nf)
    r.other_v.fill_(float("nan"))
    r.call(6, v=r.other_v, f=r.f)
    _mesh_ok(r, r.other_v, r.f)
    r.call(6, v=r.v, f=r.f)
    _mesh_ok(r, r.v, r.f)
    # 3 -> 4 -> 6, then the gather emitter on the same workspace; a starting part is legal at any time
    r.fresh().call(3)
    r.call(4)
    r.call(6, v=r.v, f=r.f)
    _mesh_ok(r, r.v, r.f)
    r.emit(r.other_v, r.f)
    r.call(3)
    r.call(0, v=r.v, f=r.f)
    r.count()
    r.emit(r.v, r.f)
    torch.cuda.synchronize()


def test_a_recycled_workspace_address_inherits_nothing(gpu):
    """The table is keyed by the workspace POINTER: the caching allocator hands the same address out again.  A starting part
    resets the entry, so the new owner's legal sequence is legal whatever the old owner did last."""
    r = Rig(gpu)
    r.call(1, split=16, v=r.v)          # ... and the extraction is abandoned
    addr = r.ws.data_ptr()
    n = r.ws.numel()
    r.ws = None
    ws2 = torch.empty(n, dtype=torch.uint8, device=gpu)
    if ws2.data_ptr() != addr:
        pytest.skip("the allocator did not hand the address out again")
    r.ws = ws2
    r.call(3)
    r.call(4)
    r.call(6, v=r.v, f=r.f)
    _mesh_ok(r, r.v, r.f)


@pytest.mark.dev_hooks
def test_the_cursor_ring_survives_a_call_that_fails_after_its_lease(gpu, built, tuning_env):
    """cursor_block_for hands a whole-grid call the stream's next pre-cleared block; the call's streaming kernel clears the
    block after it.  The ring moves on only when that kernel is enqueued (RingLease::commit): a call that fails in between
    (hook P3D_TEST_FAIL_AFTER_LEASE) must leave the following calls on clean blocks -- before round 5 `cur` advanced at the
    lease and the next call got a block nobody had cleared."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid((48, 40, 130), period=14, seed=9).to(gpu)
    ref = oracle_count(g.cpu().numpy(), 0.0)
    for _ in range(20):   # every block of the ring (16) has been used (dirty unless cleared by the call before)
        v, f = built.marching_cubes(g, 0.0)
        assert (v.shape[0], f.shape[0]) == ref
    for fails in (1, 3):
        tuning_env("P3D_TEST_FAIL_AFTER_LEASE", str(fails))
        for _ in range(fails):
            with pytest.raises(RuntimeError, match="injected failure"):
                built.marching_cubes(g, 0.0)
        for _ in range(20):   # more than a whole turn of the ring
            v, f, k = None, None, None
            v, f = built.marching_cubes(g, 0.0)
            assert (v.shape[0], f.shape[0]) == ref
            assert int(f.max()) == ref[0] - 1 and int(f.min()) == 0
        hv, hf = capi.extract_fused(g, 0.0)
        assert (hv.shape[0], hf.shape[0]) == ref
    tuning_env("P3D_TEST_FAIL_AFTER_LEASE", None)
    # whole mesh after the storm
    from tests.test_gpu_parity import _assert_same_mesh, _hip_extract_fused
    _assert_same_mesh(_hip_extract_fused(gpu, g.cpu().numpy(), 0.0, None, None), oracle_extract(g.cpu().numpy(), 0.0))


def test_the_table_holds_the_most_recent_workspaces(gpu):
    """The host-side table keeps the 1024 most recently used workspaces (the older half goes in one sweep when it is full):
    an extraction left unfinished on a workspace that fell out is 'not in progress' any more -- refused, never garbage."""
    from primitive3d_amd import capi
    r = Rig(gpu, shape=(6, 8, 70))
    n = capi.workspace_bytes(*r.shape)
    pool = [torch.empty(n, dtype=torch.uint8, device=gpu) for _ in range(1600)]
    for ws in pool:
        r.call(3, ws=ws)
    r.refuse(lambda: r.call(4, ws=pool[0]), "none is in progress")
    r.call(4, ws=pool[-1])          # the most recent ones are all there
    r.call(4, ws=pool[-500])
    torch.cuda.synchronize()


def test_part_1_keeps_its_cursor_block_while_others_start_on_the_stream(gpu):
    """Part 1 takes a pre-cleared block of the stream's ring (no clearing kernel) and HOLDS it for the parts that continue
    the streaming: other extractions that start on the stream in between -- the in-process multi-rank harness does exactly
    that -- step over held blocks, however many of them start (until round 5 the held block was overrun after 13 starts and
    the continuing part refused: ADVICE r05).  With fifteen blocks held a start keeps its cursors in its workspace header.
    Only releasing the stream's state takes a held block away: refused then, never garbage."""
    r = Rig(gpu)
    other = Rig(gpu, shape=(12, 16, 70), seed=8)
    split = 16
    r.call(1, split=split, v=r.v)
    for _ in range(40):   # two and a half turns of the ring
        other.call(0, v=other.v, f=other.f)
    _mesh_ok(other, other.v, other.f)
    r.call(3, split=split, v=r.v)
    r.call(4, v=r.v)
    assert r.capi.read_counts(r.ws) == (r.nv, r.nf)
    r.call(5, v=r.v, f=r.f)
    _mesh_ok(r, r.v, r.f)
    # twenty extractions between their part 1 and part 2 at once (world = 20 in one process): fifteen hold a ring block,
    # the rest keep their cursors in their headers; whole-grid calls run in between; every one finishes with the right mesh
    many = [Rig(gpu) for _ in range(20)]
    for m in many:
        m.call(1, split=split, v=m.v)
        other.call(0, v=other.v, f=other.f)
    for m in reversed(many):
        m.call(2, split=split, v=m.v, f=m.f)
        other.call(0, v=other.v, f=other.f)
    for m in many[::7]:
        _mesh_ok(m, m.v, m.f)
    _mesh_ok(other, other.v, other.f)
    # the blocks are free again: a whole turn of the ring by plain calls, then a held one that is taken away
    for _ in range(20):
        other.call(0, v=other.v, f=other.f)
    r.fresh().call(1, split=split, v=r.v)
    torch.cuda.synchronize()
    r.capi.release_stream(torch.cuda.current_stream().cuda_stream)
    with pytest.raises(r.capi.P3DError, match="cursor block part 1 took is gone"):
        r.call(2, split=split, v=r.v, f=r.f)
    r.call(0, v=r.v, f=r.f)   # (a start is always legal; the stream's state is created again)
    _mesh_ok(r, r.v, r.f)
    torch.cuda.synchronize()


def test_part_3_exports_the_first_planes_records_with_the_header(gpu):
    """p3d_mc_slab.export_first_plane_to: the launch that writes the header also writes plane 0's dense vertex-id records --
    the same bytes p3d_mc_export_plane_records gives as a launch of its own."""
    r = Rig(gpu)
    _, nbytes = r.capi.plane_records(r.ws, *r.shape, 0)
    out = torch.zeros(nbytes, dtype=torch.uint8, device=gpu)
    slab = r.capi.Slab()
    slab.part, slab.export_first_plane_to = 3, out.data_ptr()
    r.capi.extract_fused_raw(r.g, 0.0, r.lower, r.upper, r.ws, None, None, slab=slab, scratch=r.scratch)
    ref = r.capi.export_plane_records(r.ws, *r.shape, 0, torch.zeros_like(out))
    torch.cuda.synchronize()
    # (records of units without vertices hold garbage in both: compare the units that own vertices)
    a, b = out.view(torch.int32).view(-1, 2).cpu(), ref.view(torch.int32).view(-1, 2).cpu()
    owns = (b[:, 1] != 0) | (a[:, 1] != 0)
    bits_off = r.capi.debug_layout(*r.shape)
    assert owns.any()
    assert torch.equal(a[owns], b[owns])


def test_rank_counts_serve_64_ranks(gpu):
    r = Rig(gpu)
    rc = torch.zeros(80, dtype=torch.int64, device=gpu)
    r.call(3)
    r.call(4)
    slab = r.capi.Slab()
    slab.part, slab.rank_counts, slab.rank = 6, rc.data_ptr(), 64
    with pytest.raises(r.capi.P3DError, match="ranks 0..63"):
        r.capi.extract_fused_raw(r.g, 0.0, r.lower, r.upper, r.ws, r.v, r.f, slab=slab, scratch=r.scratch)
    # rank 63 with made-up counts of the ranks before it: the ids are shifted by their sum
    rc[:63] = 1000
    rc[63] = r.nv
    slab.rank = 63
    r.capi.extract_fused_raw(r.g, 0.0, r.lower, r.upper, r.ws, r.v, r.f, slab=slab, scratch=r.scratch)
    torch.cuda.synchronize()
    assert int(r.f.min()) == 63000 and int(r.f.max()) == 63000 + r.nv - 1
