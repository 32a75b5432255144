"""BASELINE.json configurations at their full sizes, checked through size-independent properties and
an independent on-device count (plain torch ops, no code shared with the HIP kernels)."""
import numpy as np
import pytest
import torch

from oracle import oracle_count
from oracle.np_counts import tri_counts

pytestmark = pytest.mark.gpu


def torch_counts(g, thresh, planes_per_step=64):
    """V and F as count_vertices_faces_kernel defines them (marching_cubes.cu:25-66), with torch ops; the grid is
    walked in axis-0 pieces (each with the next plane attached) so that 1024^3 fits comfortably."""
    nt = torch.from_numpy(tri_counts()).to(g.device)
    corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    v = f = 0
    for x0 in range(0, g.shape[0], planes_per_step):
        x1 = min(x0 + planes_per_step, g.shape[0])
        ins = g[x0:min(x1 + 1, g.shape[0])].float() > thresh   # planes x0..x1 (x1 = first plane of the next piece)
        own = ins[:x1 - x0]
        v += int((ins[1:] != ins[:-1]).sum() + (own[:, 1:] != own[:, :-1]).sum() + (own[:, :, 1:] != own[:, :, :-1]).sum())
        c = ins.to(torch.int16)
        sx, sy, sz = ins.shape[0] - 1, g.shape[1] - 1, g.shape[2] - 1
        if sx < 1:
            continue
        mask = torch.zeros((sx, sy, sz), dtype=torch.int16, device=g.device)
        for bit, (dx, dy, dz) in enumerate(corners):
            mask |= c[dx:dx + sx, dy:dy + sy, dz:dz + sz] << bit
        f += int(nt[mask.long()].sum())
    return v, f


def soup_hashes(v, f):
    """Order-free fingerprint of a mesh: one 64-bit hash per triangle of the BIT PATTERNS of its nine coordinates in
    corner order (winding kept), sorted.  Equal sorted hashes <=> equal triangle soups (up to 64-bit collisions)."""
    bits = v.contiguous().view(torch.int32)[f.long()].reshape(-1, 9).long() & 0xFFFFFFFF
    h = torch.zeros(bits.shape[0], dtype=torch.int64, device=v.device)
    for k in range(9):  # multiplicative mixing in wrapping int64 arithmetic
        h = (h ^ bits[:, k]) * -7046029254386353131 + (k + 1)
        h = h ^ (h >> 29)
    return torch.sort(h).values


def mesh_properties(v, f):
    assert f.min() >= 0 and f.max() < v.shape[0]
    used = torch.zeros(v.shape[0], dtype=torch.bool, device=v.device)
    used[f.flatten().long()] = True
    assert bool(used.all()), "every vertex is referenced by some triangle"
    # every directed edge is used by at most one triangle (consistent winding, no duplicated faces)
    fl = f.long()
    e = torch.cat([fl[:, [0, 1]], fl[:, [1, 2]], fl[:, [2, 0]]])
    key = e[:, 0] * v.shape[0] + e[:, 1]
    ne = e[:, 0] != e[:, 1]
    assert bool(ne.all())
    assert key.unique().numel() == key.numel()
    assert bool(torch.isfinite(v).all())


def test_c3_perlin_512_full_size(gpu, built):
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid(512, period=64, seed=0, device=gpu)
    v, f = built.marching_cubes(g, 0.0)
    ev, ef = torch_counts(g, 0.0)
    assert (v.shape[0], f.shape[0]) == (ev, ef)
    mesh_properties(v, f)
    assert float(v.min()) >= 0.0 and float(v.max()) <= 511.0
    # a second call (size hints now known) must give the same mesh up to order: compare sorted positions
    v2, f2 = built.marching_cubes(g, 0.0)
    assert v2.shape == v.shape and f2.shape == f.shape
    a = torch.sort(v.double() @ torch.tensor([1.0, 1e3, 1e6], dtype=torch.float64, device=gpu)).values
    b = torch.sort(v2.double() @ torch.tensor([1.0, 1e3, 1e6], dtype=torch.float64, device=gpu)).values
    assert torch.equal(a, b)


def test_c3_whole_mesh_equals_the_oracle(gpu, built):
    """The headline workload at its full size: the WHOLE 512^3 mesh (positions bit for bit, winding kept) against the CPU
    oracle run on all host cores -- not only counts and properties."""
    from primitive3d_amd.fields import perlin_grid
    from oracle import oracle_extract
    g = perlin_grid(512, period=64, seed=0, device=gpu)
    v, f = built.marching_cubes(g, 0.0)
    rv, rf, _ = oracle_extract(g.cpu().numpy(), 0.0, threads=0, want_keys=False)
    assert v.shape[0] == rv.shape[0] and f.shape[0] == rf.shape[0]
    assert torch.equal(soup_hashes(v, f), soup_hashes(torch.from_numpy(rv).to(gpu), torch.from_numpy(rf).to(gpu)))
    a = np.sort(v.cpu().numpy().view([("", np.float32)] * 3), axis=0)
    assert np.array_equal(a, np.sort(rv.view([("", np.float32)] * 3), axis=0))


def test_c3_fresh_grids_in_turn(gpu, built):
    """bench.py's `modes.fresh_grid` workload at its full size: four distinct 512^3 grids (seeds 0..3) taken in turn through
    the boundary call -- every call's counts against the independent count, closed-manifold properties, and the meshes of
    the second turn equal to the first turn's (the size hints come from the OTHER grids' calls: every call is one pass)."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    grids = [perlin_grid(512, period=64, seed=sd, device=gpu) for sd in range(4)]
    want = [torch_counts(g, 0.0) for g in grids]
    assert len(set(want)) == 4
    first = []
    for turn in range(2):
        for i, g in enumerate(grids):
            p0 = capi.debug_counters()["streaming_passes"]
            v, f = built.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, [512.0] * 3)
            assert (v.shape[0], f.shape[0]) == want[i], (turn, i)
            if turn == 0:
                mesh_properties(v, f)
                first.append(soup_hashes(v, f))
            else:
                assert capi.debug_counters()["streaming_passes"] - p0 == 1, (turn, i)
                assert torch.equal(soup_hashes(v, f), first[i]), (turn, i)


def test_c3_four_octave_field_whole_mesh(gpu, built):
    """SURVEY.md 8d, C3's secondary workload at full size: four octaves (period 64 -> 8, persistence 0.5), about twice the
    surface of the single-octave field and much finer detail (more vertices per wave-plane, denser face tiles).  Whole mesh
    against the oracle on all host cores, positions bit for bit."""
    from primitive3d_amd.fields import perlin_grid
    from oracle import oracle_extract
    g = perlin_grid(512, period=64, seed=0, octaves=4, persistence=0.5, device=gpu)
    v, f = built.marching_cubes(g, 0.0)
    assert (v.shape[0], f.shape[0]) == torch_counts(g, 0.0) and v.shape[0] > 8_000_000
    rv, rf, _ = oracle_extract(g.cpu().numpy(), 0.0, threads=0, want_keys=False)
    assert v.shape[0] == rv.shape[0] and f.shape[0] == rf.shape[0]
    assert torch.equal(soup_hashes(v, f), soup_hashes(torch.from_numpy(rv).to(gpu), torch.from_numpy(rf).to(gpu)))
    a = np.sort(v.cpu().numpy().view([("", np.float32)] * 3), axis=0)
    assert np.array_equal(a, np.sort(rv.view([("", np.float32)] * 3), axis=0))


def test_c2_bunny_resampled_256(gpu, built):
    """bunny.npy (66^3) trilinearly resampled to 256^3 as SURVEY.md section 8d defines C2; native 66^3 counts too."""
    from pathlib import Path
    b = torch.from_numpy(np.load(Path(__file__).parent / "golden" / "bunny66.npy"))
    v, f = built.marching_cubes(b.to(gpu), 0.0)
    assert (v.shape[0], f.shape[0]) == (13282, 26560)
    big = torch.nn.functional.interpolate(b[None, None], size=(256,) * 3, mode="trilinear", align_corners=True)[0, 0]
    v, f = built.marching_cubes(big.to(gpu), 0.0)
    assert (v.shape[0], f.shape[0]) == oracle_count(big.numpy(), 0.0) == torch_counts(big.to(gpu), 0.0)
    mesh_properties(v, f)
    from oracle import oracle_extract
    rv, rf, _ = oracle_extract(big.numpy(), 0.0, threads=0, want_keys=False)   # the whole 256^3 mesh, bit for bit
    assert torch.equal(soup_hashes(v, f), soup_hashes(torch.from_numpy(rv).to(gpu), torch.from_numpy(rf).to(gpu)))
    # the bunny SDF is closed inside the grid: Euler characteristic 2
    e = torch.cat([f.long()[:, [0, 1]], f.long()[:, [1, 2]], f.long()[:, [2, 0]]]).sort(dim=1).values
    n_e = (e[:, 0] * v.shape[0] + e[:, 1]).unique().numel()
    assert v.shape[0] - n_e + f.shape[0] == 2


def test_c5_batched_fp16(gpu, built):
    """config 5 at reduced batch for the oracle comparison, full item size property check for one item"""
    from primitive3d_amd.fields import perlin_grid
    from oracle import canonical_mesh, oracle_extract  # noqa: F401
    grids = torch.stack([perlin_grid(48, period=16, seed=s) for s in range(3)]).half()
    v, f, vo, fo = built.marching_cubes_batched(grids.to(gpu), 0.0)
    assert vo.shape == (4,) and fo.shape == (4,) and int(vo[-1]) == v.shape[0] and int(fo[-1]) == f.shape[0]
    for b in range(3):
        ev, ef = oracle_count(grids[b].float().numpy(), 0.0)
        assert int(vo[b + 1] - vo[b]) == ev and int(fo[b + 1] - fo[b]) == ef
        fb = f[fo[b]:fo[b + 1]]
        assert fb.numel() == 0 or (int(fb.min()) >= 0 and int(fb.max()) < ev)
    big = perlin_grid(256, period=64, seed=7, device=gpu).half()
    v, f, vo, fo = built.marching_cubes_batched(big[None], 0.0)
    assert (v.shape[0], f.shape[0]) == torch_counts(big, 0.0)
    mesh_properties(v, f)


@pytest.mark.parametrize("shape,B,dtype", [((7, 9, 70), 5, torch.float32), ((33, 17, 200), 3, torch.float16),
                                           ((2, 2, 2), 9, torch.float32), ((12, 40, 517), 2, torch.float32),
                                           ((24, 24, 24), 1, torch.float16)])
def test_batched_call_equals_per_item_oracle(gpu, built, shape, B, dtype):
    """The one-launch batched entry on odd shapes (ragged rz, tiny items, a single item): every item's mesh == the
    oracle's mesh of that item alone (triangle soup, bit for bit), face ids local to the item."""
    from oracle import oracle_extract
    rng = np.random.default_rng(sum(shape) + B)
    grids = torch.from_numpy(rng.standard_normal((B,) + shape).astype(np.float32)).to(dtype)
    lower, upper = [0.5, -1.0, 2.0], [3.0, 4.0, 9.0]
    v, f, vo, fo = built.marching_cubes_batched(grids.to(gpu), 0.1, scale=(lower, upper))
    torch.cuda.synchronize()
    assert int(vo[0]) == 0 and int(fo[0]) == 0 and int(vo[-1]) == v.shape[0] and int(fo[-1]) == f.shape[0]
    soup = lambda vv, ff: np.sort(vv[ff.astype(np.int64)].reshape(len(ff), 9).view([("", np.float32)] * 9), axis=0)
    for b in range(B):
        rv, rf, _ = oracle_extract(grids[b].float().numpy(), 0.1, lower, upper)
        vb, fb = v[vo[b]:vo[b + 1]].cpu().numpy(), f[fo[b]:fo[b + 1]].cpu().numpy()
        assert vb.shape == rv.shape and fb.shape == rf.shape, (b, vb.shape, rv.shape, fb.shape, rf.shape)
        assert fb.size == 0 or (fb.min() >= 0 and fb.max() < vb.shape[0])
        assert np.array_equal(soup(vb, fb), soup(rv, rf)), b
    # a second call takes the size hints of the first
    v2, f2, vo2, fo2 = built.marching_cubes_batched(grids.to(gpu), 0.1, scale=(lower, upper))
    assert torch.equal(vo2, vo) and torch.equal(fo2, fo)


@pytest.mark.dev_hooks
def test_batched_call_falls_back_item_by_item(gpu, built, tuning_env):
    """P3D_TEST_ID_LIMIT makes every region report an id-space overflow: the wrapper must return the same meshes
    through its per-item path."""
    from oracle import oracle_count
    grids = torch.from_numpy(np.random.default_rng(3).standard_normal((3, 10, 11, 70)).astype(np.float32))
    tuning_env("P3D_TEST_ID_LIMIT", "4")
    v, f, vo, fo = built.marching_cubes_batched(grids.to(gpu), 0.0)
    for b in range(3):
        assert (int(vo[b + 1] - vo[b]), int(fo[b + 1] - fo[b])) == oracle_count(grids[b].numpy(), 0.0)


@pytest.mark.dev_hooks
@pytest.mark.parametrize("kind", ["stack", "single"])
def test_chunk_prefix_handoff_under_load(gpu, built, tuning_env, kind):
    """With more than 1024 face chunks a one-block scan between the counting and the face launch (k_chunk_prefix; for a
    stack of items inside k_stack_finish) turns the chunk totals into their exclusive prefix, which every face tile then
    reads instead of adding the totals in front of its chunk up (p3d_mc.hip).  A wrong prefix would silently shift every
    face offset behind it.  Reference = the same call with P3D_NO_CHUNK_PRE=1 (every face tile adds the totals up itself):
    the triangle soup must be the same bit for bit (vertex ids are arrival order, so the soups are compared), counts equal
    to the independent torch count -- repeated, back to back, on a loaded chip with warm caches.  (Rounds 2-3 had the last
    counting block make the prefix inside the launch, with a fence-free hand-off; that is gone, the test stayed.)"""
    from primitive3d_amd.fields import perlin_grid
    if kind == "stack":   # 20 items x 68 chunks = 1360 chunks
        grids = torch.stack([perlin_grid((130, 256, 256), period=32, seed=s, device=gpu).half() for s in range(20)])
        expect = [sum(x) for x in zip(*[torch_counts(grids[b], 0.0) for b in range(20)])]

        def run():   # (face ids are local to an item: one soup per item, concatenated)
            v, f, vo, fo = built.marching_cubes_batched(grids, 0.0)
            vo, fo = vo.tolist(), fo.tolist()
            return v.shape[0], f.shape[0], torch.cat([soup_hashes(v[vo[b]:vo[b + 1]], f[fo[b]:fo[b + 1]]) for b in range(20)])
    else:                 # 65 x 16 = 1040 chunks
        g = perlin_grid((520, 512, 512), period=48, seed=5, device=gpu).half()
        from primitive3d_amd import capi
        expect = list(torch_counts(g, 0.0))

        def run():
            v, f = capi.extract_fused(g, 0.0, cap_vertices=1 << 24, cap_faces=1 << 25)
            return v.shape[0], f.shape[0], soup_hashes(v, f)
    tuning_env("P3D_NO_CHUNK_PRE", "1")
    nv0, nf0, h0 = run()
    torch.cuda.synchronize()
    assert [nv0, nf0] == expect
    tuning_env("P3D_NO_CHUNK_PRE", None)
    for _ in range(12):
        nv, nf, h = run()
        assert (nv, nf) == (nv0, nf0)
        assert torch.equal(h, h0)


@pytest.mark.dev_hooks
def test_batch_whose_totals_exceed_int32_goes_item_by_item(gpu, built, tuning_env):
    """Face ids are local to an item, so only an ITEM is bound by int32 -- but the one-launch path reports the batch
    totals through p3d_mc_read_counts, which refuses totals beyond int32 (P3D_ERANGE).  P3D_TEST_INDEX_LIMIT pretends
    that limit is 6000: the batch (3 items of ~4400 faces) exceeds it, every item fits."""
    grids = torch.from_numpy(np.random.default_rng(8).standard_normal((3, 9, 10, 20)).astype(np.float32))
    counts = [oracle_count(grids[b].numpy(), 0.0) for b in range(3)]
    assert max(c[1] for c in counts) < 6000 < sum(c[1] for c in counts)
    from primitive3d_amd import capi
    import importlib
    mcmod = importlib.import_module("primitive3d_amd.marching_cubes")   # (the package exports the FUNCTION of that name)
    tuning_env("P3D_TEST_INDEX_LIMIT", "6000")
    key = (gpu.index, 3, 9, 10, 20)
    try:
        passes = []
        ncalls = mcmod._PER_ITEM_CALLS + 2
        for call in range(ncalls):
            before = capi.debug_counters()["streaming_passes"]
            v, f, vo, fo = built.marching_cubes_batched(grids.to(gpu), 0.0)
            passes.append(capi.debug_counters()["streaming_passes"] - before)
            for b in range(3):
                assert (int(vo[b + 1] - vo[b]), int(fo[b + 1] - fo[b])) == counts[b]
                fb = f[fo[b]:fo[b + 1]]
                assert int(fb.min()) >= 0 and int(fb.max()) < counts[b][0]
            if call == 0:
                assert mcmod._BATCH_HINTS.get(key) == ("per_item", mcmod._PER_ITEM_CALLS)
        # the first call streams the whole batch once before it learns of the limit; the next ones go item by item at once
        # (it used to allocate and stream the batch again on every call) -- until the marker has expired: the
        # one-launch path is then tried again (it used to stay for the life of the process), here only to learn
        # of the limit once more
        assert passes[0] > passes[1] >= 3, passes
        assert passes[1:mcmod._PER_ITEM_CALLS + 1] == [passes[1]] * mcmod._PER_ITEM_CALLS, passes
        assert passes[mcmod._PER_ITEM_CALLS + 1] == passes[0], passes
    finally:
        mcmod._BATCH_HINTS.pop(key, None)


def test_c5_32x256_fp16_full_batch(gpu, built):
    """BASELINE.json configs[4] at its stated size: 32 x 256^3 fp16 density grids through marching_cubes_batched.
    Every item: counts against the independent torch count + mesh properties; four items: the whole mesh (triangle
    soup, positions bit-exact, winding kept) against the oracle applied to grid[b].float()."""
    from primitive3d_amd.fields import perlin_grid
    from oracle import oracle_extract
    B = 32
    grids = torch.stack([perlin_grid(256, period=64, seed=s, device=gpu).half() for s in range(B)])
    v, f, vo, fo = built.marching_cubes_batched(grids, 0.0)
    torch.cuda.synchronize()
    assert vo.shape == (B + 1,) and fo.shape == (B + 1,) and int(vo[-1]) == v.shape[0] and int(fo[-1]) == f.shape[0]
    assert v.dtype == torch.float32 and f.dtype == torch.int32
    for b in range(B):
        vb, fb = v[vo[b]:vo[b + 1]], f[fo[b]:fo[b + 1]]
        assert (vb.shape[0], fb.shape[0]) == torch_counts(grids[b], 0.0), b
        mesh_properties(vb, fb)
    for b in (0, 7, 19, 31):
        vb, fb = v[vo[b]:vo[b + 1]], f[fo[b]:fo[b + 1]]
        rv, rf, _ = oracle_extract(grids[b].float().cpu().numpy(), 0.0, threads=0, want_keys=False)
        assert vb.shape[0] == rv.shape[0] and fb.shape[0] == rf.shape[0]
        ref = soup_hashes(torch.from_numpy(rv).to(gpu), torch.from_numpy(rf).to(gpu))
        assert torch.equal(soup_hashes(vb, fb), ref), f"item {b}: triangle soup differs from the oracle's"
        # and literally, not only by hash, on the sorted vertex positions
        a = np.sort(vb.cpu().numpy().view([("", np.float32)] * 3), axis=0)
        assert np.array_equal(a, np.sort(rv.view([("", np.float32)] * 3), axis=0))


def test_c4_1024_cubed_on_one_gpu(gpu, built):
    """BASELINE.json configs[3]'s volume on ONE GPU (the baseline its >= 6x target is defined against, and the size
    where the reference's int32 x*(res_y*res_z*3), marching_cubes.cu:98,161, overflows): counts against the torch
    count, mesh properties, then the same volume as 8 axis-0 slabs run in this process -- merged counts, global ids
    and the whole triangle soup equal to the plain call's."""
    from primitive3d_amd.fields import perlin_grid
    from primitive3d_amd.slab import extract_in_process
    g = perlin_grid(1024, period=64, seed=0, device=gpu)
    v, f = built.marching_cubes(g, 0.0)
    torch.cuda.synchronize()
    assert (v.shape[0], f.shape[0]) == torch_counts(g, 0.0)
    assert 3 * g.numel() > 2 ** 31, "this is the size the reference cannot index"
    mesh_properties(v, f)
    assert float(v.min()) >= 0.0 and float(v.max()) <= 1023.0
    whole = soup_hashes(v, f)
    nv, nf = v.shape[0], f.shape[0]
    del v, f
    res = extract_in_process(g, 8, 0.0)
    torch.cuda.synchronize()
    assert sum(r.vertices.shape[0] for r in res) == nv and sum(r.faces.shape[0] for r in res) == nf
    bases = [r.vertex_base for r in res]
    assert bases == [sum(r.vertices.shape[0] for r in res[:i]) for i in range(8)]
    vm = torch.cat([r.vertices for r in res])
    fm = torch.cat([r.faces for r in res])
    for i, r in enumerate(res):  # a slab's vertices lie in its own x range
        assert float(r.vertices[:, 0].min()) >= 128.0 * i and float(r.vertices[:, 0].max()) <= 128.0 * (i + 1)
    assert torch.equal(soup_hashes(vm, fm), whole), "8-slab mesh differs from the single call's"


def test_c1_sphere64_through_wrapper(gpu, built):
    from primitive3d_amd.fields import sphere_grid
    v, f = built.marching_cubes(sphere_grid(64), 0)  # int64 ndarray, like examples/sphere.py
    assert (v.shape[0], f.shape[0]) == (1182, 2360)


def test_c4_shape_two_slabs_of_1024_squared(gpu):
    """config 4's per-rank shape (128 planes of 1024^2 + halo) for two neighbouring ranks, run in one process:
    counts against the independent torch count of the 256x1024x1024 grid, global face ids in range, and the
    vertices both slabs produce lie in their own x ranges."""
    from primitive3d_amd.fields import perlin_grid
    from primitive3d_amd.slab import extract_in_process
    g = perlin_grid((256, 1024, 1024), period=64, seed=0, device=gpu)
    res = extract_in_process(g, 2, 0.0)
    torch.cuda.synchronize()
    ev, ef = torch_counts(g, 0.0)
    assert sum(r.vertices.shape[0] for r in res) == ev and sum(r.faces.shape[0] for r in res) == ef
    v = torch.cat([r.vertices for r in res])
    f = torch.cat([r.faces for r in res])
    mesh_properties(v, f)
    assert float(res[0].vertices[:, 0].max()) < 128.0 and float(res[1].vertices[:, 0].min()) >= 128.0


def test_grid_above_4gib(gpu, built):
    """1280 x 1024 x 1024 fp32 = 5 GiB: byte offsets into the grid exceed 32 bits (the streaming kernel re-bases its
    buffer descriptor per plane) and 3 * voxels is 2 x the reference's int32 index range (marching_cubes.cu:98,161).
    Counts against the independent torch count, mesh properties, and the last planes really contribute."""
    from primitive3d_amd.fields import perlin_grid
    shape = (1280, 1024, 1024)
    g = perlin_grid(shape, period=64, seed=1, device=gpu)
    assert g.numel() * 4 > 2 ** 32
    v, f = built.marching_cubes(g, 0.0)
    torch.cuda.synchronize()
    assert (v.shape[0], f.shape[0]) == torch_counts(g, 0.0)
    mesh_properties(v, f)
    assert float(v[:, 0].max()) > 1270.0 and float(v[:, 0].min()) < 1.0
    # the part beyond the 4 GiB mark alone, as its own grid: same triangles as the whole grid's there (shifted in x)
    x0 = 1100
    vt, ft = built.marching_cubes(g[x0:].contiguous(), 0.0)
    sel = (v[f.long()][:, :, 0].min(dim=1).values >= x0)   # triangles of the whole mesh with all corners at x >= x0
    assert int(sel.sum()) == ft.shape[0]
