"""The predicted region layout of the one-pass call (p3d_mc_slab.region_first_rows, include/p3d_mc.h): the streaming kernel
stores every vertex at its region's first row + slot inside the CALLER's vertex buffer; a wave-plane that does not fit into
what is left of its region takes rows of the spill area behind the regions.  Rows are final unless they lie at or beyond V;
those few are moved into the free rows below V and the face kernel translates their ids.  No scratch, no second trip.
Whole meshes against the oracle (keys rebuilt from the workspace: tests/ws_keys.py also checks the header's tail tables
against their definition and that every id is assigned once), for layouts that predict every region exactly, too few rows
(spills), too many (holes; the top of the last region beyond V), both, none at all for some regions, the totals of another
field, and a spill area that is too small (flag 4 -> p3d_mc_emit)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import canonical_mesh, oracle_extract

pytestmark = pytest.mark.gpu


def _mesh(capi, ws, shape, v, f):
    from tests.ws_keys import vertex_keys_from_workspace
    torch.cuda.synchronize()
    keys = vertex_keys_from_workspace(ws.cpu().numpy(), shape, v.shape[0], capi.debug_layout(*shape))
    return canonical_mesh(v.cpu().numpy(), f.cpu().numpy(), keys)


def _same(a, b):
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True)


def _first_call(capi, g, thresh, lower, upper):
    """A scratch-mode call: the mesh's size and the region totals a caller would lay the next call out from."""
    ws = torch.empty(capi.workspace_bytes(*g.shape), dtype=torch.uint8, device=g.device)
    capv = max(4096, 2 * g.numel())   # (white noise: 1.5 vertices per voxel)
    v = torch.empty((capv, 3), device=g.device)
    f = torch.empty((2 * capv, 3), dtype=torch.int32, device=g.device)
    capi.extract_fused_raw(g, thresh, lower, upper, ws, v, f)
    nv, nf, flags, regions = capi.read_counts_ex(ws)
    assert flags == 0 and sum(regions) == nv
    return nv, nf, regions


def _layout_call(capi, g, thresh, lower, upper, first, rows, nf_cap, guard=0):
    ws = torch.empty(capi.workspace_bytes(*g.shape), dtype=torch.uint8, device=g.device)
    v = torch.full((rows + guard, 3), -7.0, device=g.device)
    f = torch.full((max(1, nf_cap), 3), -1, dtype=torch.int32, device=g.device)
    slab = capi.Slab()
    slab.region_first_rows = ctypes.cast(first, ctypes.c_void_p)
    capi.extract_fused_raw(g, thresh, lower, upper, ws, v[:rows], f, slab=slab)   # (no scratch buffer)
    nv, nf, flags, regions = capi.read_counts_ex(ws)
    return ws, v, f, nv, nf, flags, regions


CASES = [((40, 50, 517), "perlin"), ((33, 30, 1100), "perlin"), ((130, 131, 200), "perlin"), ((24, 40, 512), "noise"),
         ((9, 100, 129), "perlin"), ((200, 96, 512), "perlin")]


def _prediction(kind, regions):
    """(extra rows per region relative to what it needs, rows of the eight spill areas together: an eighth each)"""
    n = [int(r) for r in regions]
    if kind == "exact":
        return [0] * 32, 64
    if kind == "under":     # every third region is given a tenth less than it needs: its last wave-planes spill
        return [-(v // 10) if r % 3 == 0 else 0 for r, v in enumerate(n)], sum(n) // 2 + 4096
    if kind == "over":      # every region a seventh more: holes behind all of them, the top regions lie beyond V
        return [v // 7 + 5 for v in n], 256
    if kind == "mixed":
        return [(-(v // 6) if r % 2 else v // 9 + 3) for r, v in enumerate(n)], sum(n) + 4096
    if kind == "none_for_some":   # regions 4..11 are given no rows at all: everything of theirs spills
        return [-v if 4 <= r < 12 else 0 for r, v in enumerate(n)], 2 * sum(n) + 4096   # (areas 1 and 2 take it all: an eighth each)
    raise KeyError(kind)


@pytest.mark.parametrize("shape,kind", CASES)
@pytest.mark.parametrize("pred", ["exact", "under", "over", "mixed", "none_for_some"])
def test_layout_from_the_same_fields_totals(gpu, shape, kind, pred):
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    g = (torch.from_numpy(np.random.default_rng(3).standard_normal(shape).astype(np.float32)) if kind == "noise"
         else perlin_grid(shape, period=14, seed=sum(shape))).to(gpu)
    lower, upper = [0.5, -1.0, 2.0], [3.0, 4.0, 9.0]
    ref = canonical_mesh(*oracle_extract(g.cpu().numpy(), 0.03, lower, upper))
    nv, nf, regions = _first_call(capi, g, 0.03, lower, upper)
    assert nv == ref[0].shape[0]
    extra, spill = _prediction(pred, regions)
    first, rows = capi.region_layout(regions, spill, extra)
    ws, v, f, nv2, nf2, flags, regions2 = _layout_call(capi, g, 0.03, lower, upper, first, rows, nf, guard=32)
    assert (nv2, nf2, flags) == (nv, nf, 0) and regions2 == regions
    assert (v[rows:] == -7.0).all()   # nothing past the buffer it was given
    _same(_mesh(capi, ws, shape, v[:nv], f[:nf]), ref)
    assert int(f[:nf].max()) == nv - 1 and int(f[:nf].min()) == 0
    if pred == "exact":   # every row is where it stays: nothing was moved, no id translated
        hdr = ws[:8192].view(torch.int64).cpu()
        assert int(hdr[850 + 40]) == 0   # H_TAIL_HP[40]: free rows below V


def test_layout_from_another_fields_totals(gpu):
    """Per-frame extraction of a changing field: the layout comes from the frame before (other noise seed), the regions differ
    by a few per cent, the spill area takes it: flags 0, the whole mesh right; frames in turn, each laid out from its predecessor."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shape = (96, 120, 512)
    _, _, regions = _first_call(capi, perlin_grid(shape, period=24, seed=1).to(gpu), 0.0, [0.0] * 3, [1.0] * 3)
    for seed in (2, 3, 4):
        g = perlin_grid(shape, period=24, seed=seed).to(gpu)
        ref = canonical_mesh(*oracle_extract(g.cpu().numpy(), 0.0, [0.0] * 3, [1.0] * 3))
        first, rows = capi.region_layout(regions)   # (exact regions + a tenth of the total as the spill area)
        ws, v, f, nv, nf, flags, regions = _layout_call(capi, g, 0.0, [0.0] * 3, [1.0] * 3, first, rows, 3 * ref[0].shape[0])
        assert flags == 0 and nv == ref[0].shape[0]
        _same(_mesh(capi, ws, shape, v[:nv], f[:nf]), ref)


def test_a_spill_area_that_is_too_small_is_flagged_and_the_mesh_re_emitted(gpu):
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shape = (64, 72, 512)
    g = perlin_grid(shape, period=14, seed=5).to(gpu)
    lower, upper = [0.0] * 3, [float(s) for s in shape]
    ref = canonical_mesh(*oracle_extract(g.cpu().numpy(), 0.0, lower, upper))
    nv, nf, regions = _first_call(capi, g, 0.0, lower, upper)
    extra = [-(int(n) // 2) if r == 7 else 0 for r, n in enumerate(regions)]   # region 7 gets half the rows it needs ...
    first, rows = capi.region_layout(regions, 100, extra)                      # ... and the spill area 100 rows
    ws, v, f, nv2, nf2, flags, regions2 = _layout_call(capi, g, 0.0, lower, upper, first, rows, nf)
    assert flags == 4 and (nv2, nf2) == (nv, nf) and regions2 == regions   # the COUNTS are right; the vertex buffer is not
    v2 = torch.empty((nv, 3), device=gpu)
    f2 = torch.empty((nf, 3), dtype=torch.int32, device=gpu)
    capi.emit(g, 0.0, lower, upper, ws, v2, f2)   # a second pass, every region at its final rows
    _same(_mesh(capi, ws, shape, v2, f2), ref)
    # and the order check: behind a layout call there is no scratch to copy from (part 6), and p3d_mc_emit takes both buffers
    ws3, v3, f3, *_ = _layout_call(capi, g, 0.0, lower, upper, *capi.region_layout(regions, 64), nf)
    with pytest.raises(capi.P3DError, match="both buffers"):
        capi.emit(g, 0.0, lower, upper, ws3, v2, None)
    with pytest.raises(capi.P3DError, match="both buffers"):
        capi.emit(g, 0.0, lower, upper, ws3, None, f2)
    slab = capi.Slab()
    slab.part = 6
    with pytest.raises(capi.P3DError, match="had none"):
        capi.extract_fused_raw(g, 0.0, lower, upper, ws3, v2, f2, slab=slab, scratch=torch.empty((64, 3), device=gpu))


def test_bad_layouts_are_refused(gpu):
    from primitive3d_amd import capi
    g = torch.zeros((8, 8, 70), device=gpu)
    ws = torch.empty(capi.workspace_bytes(8, 8, 70), dtype=torch.uint8, device=gpu)
    v = torch.empty((4096, 3), device=gpu)
    f = torch.empty((4096, 3), dtype=torch.int32, device=gpu)
    for rows in ([1] + list(range(1, 41)), list(range(0, 41 * 200, 200)), [0] + list(range(60, 20, -1))):   # first != 0; beyond the buffer; descending
        slab = capi.Slab()
        arr = (ctypes.c_uint32 * 41)(*rows)
        slab.region_first_rows = ctypes.cast(arr, ctypes.c_void_p)
        with pytest.raises(capi.P3DError, match="region_first_rows"):
            capi.extract_fused_raw(g, 0.5, [0.0] * 3, [1.0] * 3, ws, v, f, slab=slab)
    # a region of more than 2^28 rows (the kernel addresses a row by a 32-bit byte offset from its region's first row)
    big = torch.empty(((1 << 28) + 4096, 3), device=gpu)   # (3.2 GB: the table must fit into the buffer to get that far)
    rows = [0] + [(1 << 28) + 1 + 8 * r for r in range(40)]
    slab = capi.Slab()
    arr = (ctypes.c_uint32 * 41)(*rows)
    slab.region_first_rows = ctypes.cast(arr, ctypes.c_void_p)
    with pytest.raises(capi.P3DError, match="2\\^28 rows"):
        capi.extract_fused_raw(g, 0.5, [0.0] * 3, [1.0] * 3, ws, big, f, slab=slab)
