"""GPU parity: HIP path (through the C ABI and through the pybind module) vs the CPU oracle, on the
canonical form (vertices keyed by edge key, faces as ordered triples of edge keys).  Bit-exact for
topology; vertex positions are required to be bit-identical too (the 1e-5 tolerance BASELINE.json
states is the outer bound; we assert equality and report the max abs diff on failure)."""
import numpy as np
import pytest
import torch

from oracle import canonical_mesh, oracle_count, oracle_extract
from tests.cases import small_cases

pytestmark = pytest.mark.gpu
TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 on interpolated fp32 vertex positions"


def _hip_extract(gpu, g, thresh, lower, upper, dtype=torch.float32):
    from primitive3d_amd import capi
    t = torch.from_numpy(np.ascontiguousarray(g)).to(gpu).to(dtype)
    v, f, k = capi.extract(t, thresh, lower, upper, with_keys=True)
    torch.cuda.synchronize()
    return v.cpu().numpy(), f.cpu().numpy(), k.cpu().numpy()


def _hip_extract_fused(gpu, g, thresh, lower, upper, dtype=torch.float32, **caps):
    """One-pass kernel; vertex keys rebuilt on the host from the workspace (tests/ws_keys.py)."""
    from primitive3d_amd import capi
    from tests.ws_keys import vertex_keys_from_workspace
    t = torch.from_numpy(np.ascontiguousarray(g)).to(gpu).to(dtype)
    v, f, ws = capi.extract_fused(t, thresh, lower, upper, return_ws=True, **caps)
    torch.cuda.synchronize()
    keys = vertex_keys_from_workspace(ws.cpu().numpy(), t.shape, v.shape[0], capi.debug_layout(*t.shape))
    return v.cpu().numpy(), f.cpu().numpy(), keys


def _assert_same_mesh(hip, ref):
    hk, hv, hf = canonical_mesh(*hip)
    rk, rv, rf = canonical_mesh(*ref)
    assert hk.shape == rk.shape and hf.shape == rf.shape, (hk.shape, rk.shape, hf.shape, rf.shape)
    assert np.array_equal(hk, rk), "vertex edge-key sets differ"
    assert np.array_equal(hf, rf), "face (edge-key triple) multisets differ"
    same = (hv == rv) | (np.isnan(hv) & np.isnan(rv))
    if not same.all():
        d = np.nanmax(np.abs(hv.astype(np.float64) - rv.astype(np.float64)))
        assert d <= TOL, f"vertex positions differ by {d}"
        pytest.fail(f"vertex positions within tolerance ({d}) but not bit-identical")


@pytest.mark.parametrize("name", sorted(small_cases().keys()))
def test_small_cases_match_oracle(gpu, name):
    g, thresh, lower, upper = small_cases()[name]
    hip = _hip_extract(gpu, g, thresh, lower, upper)
    ref = oracle_extract(g, thresh, lower, upper)
    _assert_same_mesh(hip, ref)
    # ids are a permutation-free dense range
    if hip[1].size:
        assert hip[1].min() >= 0 and hip[1].max() < hip[0].shape[0]


def _hip_extract_pair(gpu, g, thresh, lower, upper, dtype=torch.float32):
    """The literal two-phase binding: p3d_mc_count -> read (V, F) -> exactly sized tensors -> p3d_mc_emit,
    i.e. the one-pass kernels in count-only form and a second streaming pass that stores every region at its final rows;
    vertex keys rebuilt on the host from the workspace (tests/ws_keys.py)."""
    from primitive3d_amd import capi
    from tests.ws_keys import vertex_keys_from_workspace
    t = torch.from_numpy(np.ascontiguousarray(g)).to(gpu).to(dtype)
    v, f, ws = capi.extract(t, thresh, lower, upper, return_ws=True)
    torch.cuda.synchronize()
    keys = vertex_keys_from_workspace(ws.cpu().numpy(), t.shape, v.shape[0], capi.debug_layout(*t.shape))
    return v.cpu().numpy(), f.cpu().numpy(), keys


@pytest.mark.parametrize("name", sorted(small_cases().keys()))
def test_small_cases_count_emit_pair_match_oracle(gpu, name):
    g, thresh, lower, upper = small_cases()[name]
    _assert_same_mesh(_hip_extract_pair(gpu, g, thresh, lower, upper), oracle_extract(g, thresh, lower, upper))


@pytest.mark.parametrize("shape,dtype", [((40, 50, 517), torch.float32), ((33, 30, 1100), torch.float32),
                                         ((70, 200, 256), torch.float16), ((9, 100, 129), torch.float32),
                                         ((130, 131, 200), torch.float32), ((24, 40, 512), torch.float32)])
def test_count_emit_pair_on_the_tile_geometries(gpu, shape, dtype):
    """Every tile geometry of the streaming kernel (8 x 3, 4 x 6 / 4 x 3, 2 x 15, rows split over two launches, two z tiles,
    enough blocks for the fixed cursor groups and few enough for the rotating ones): the second pass must reproduce the
    first pass's 32 region totals exactly, or vertices land in a neighbour's rows -- the whole mesh against the oracle,
    on Perlin noise and (last shape) on white noise."""
    from primitive3d_amd.fields import perlin_grid
    if shape == (24, 40, 512):
        g = np.random.default_rng(3).standard_normal(shape).astype(np.float32)
    else:
        g = perlin_grid(shape, period=14, seed=sum(shape)).numpy()
    if dtype == torch.float16:
        g = g.astype(np.float16)
    ref = oracle_extract(g.astype(np.float32), 0.03)
    _assert_same_mesh(_hip_extract_pair(gpu, g, 0.03, None, None, dtype=dtype), ref)


@pytest.mark.parametrize("name", ["noise_33x17x200", "perlin48", "noise_9x5x129"])
def test_counting_pass_alone_leaves_consistent_records(gpu, name):
    """p3d_mc_count makes no vertex, but its records are what a slab's gather emitter and the record export read: every id
    in [0, V) must be assigned to exactly one crossing edge (tests/ws_keys.py checks that while it rebuilds the keys), and
    the keys must be the oracle's."""
    from primitive3d_amd import capi
    from tests.ws_keys import vertex_keys_from_workspace
    g, thresh, lower, upper = small_cases()[name]
    t = torch.from_numpy(np.ascontiguousarray(g)).to(gpu)
    ws = torch.empty(capi.workspace_bytes(*t.shape), dtype=torch.uint8, device=gpu)
    capi.count(t, thresh, ws)
    nv, nf = capi.read_counts(ws)
    torch.cuda.synchronize()
    keys = vertex_keys_from_workspace(ws.cpu().numpy(), t.shape, nv, capi.debug_layout(*t.shape))
    rk = oracle_extract(g, thresh, lower, upper)[2]
    assert (nv, nf) == oracle_count(g, thresh) and np.array_equal(np.sort(keys), np.sort(rk))


def test_count_emit_pair_tolerates_a_too_small_vertex_buffer(gpu):
    """p3d_mc_emit writes nothing past the capacities it is given (include/p3d_mc.h): a vertex buffer shorter than V gets
    the rows that fit -- regions are clipped at the end of the buffer, the guard rows behind it stay untouched."""
    from primitive3d_amd import capi
    g, thresh, lower, upper = small_cases()["noise_33x17x200"]
    t = torch.from_numpy(g).to(gpu)
    lower = [0.0] * 3 if lower is None else lower
    upper = [float(n) for n in t.shape] if upper is None else upper
    ws = torch.empty(capi.workspace_bytes(*t.shape), dtype=torch.uint8, device=gpu)
    capi.count(t, thresh, ws)
    nv, nf = capi.read_counts(ws)
    short = nv // 2
    v = torch.full((short + 64, 3), -7.0, device=gpu)
    f = torch.full((nf, 3), -1, dtype=torch.int32, device=gpu)
    capi.emit(t, thresh, lower, upper, ws, v[:short], f)
    torch.cuda.synchronize()
    assert (v[short:] == -7.0).all() and int(f.min()) >= 0 and int(f.max()) < nv


@pytest.mark.parametrize("name", sorted(small_cases().keys()))
def test_small_cases_fused_match_oracle(gpu, name):
    g, thresh, lower, upper = small_cases()[name]
    hip = _hip_extract_fused(gpu, g, thresh, lower, upper)
    ref = oracle_extract(g, thresh, lower, upper)
    _assert_same_mesh(hip, ref)


@pytest.mark.parametrize("name", ["noise_33x17x200", "noise_5x7x9_box", "sphere32", "all_outside_4x4x4", "noise_9x5x129"])
def test_count_allocate_emit_over_one_pass_through_the_c_abi(gpu, name):
    """The reference's order -- count, read (V, F), allocate exactly, emit (marching_cubes.cu:242-287) -- over ONE pass of
    the field: p3d_mc_extract_fused part 3 (stream into the scratch), part 4 (count the faces, totals to the host), then
    buffers of exactly V and F rows, part 6 (faces + the whole vertex compaction into them).  include/p3d_mc.h."""
    from primitive3d_amd import capi
    from tests.ws_keys import vertex_keys_from_workspace
    g, thresh, lower, upper = small_cases()[name]
    t = torch.from_numpy(np.ascontiguousarray(g)).to(gpu).float()
    ref = oracle_extract(g, thresh, lower, upper)
    lower = [0.0, 0.0, 0.0] if lower is None else lower
    upper = [float(n) for n in t.shape] if upper is None else upper
    ws = torch.empty(capi.workspace_bytes(*t.shape), dtype=torch.uint8, device=gpu)
    scratch = torch.empty((capi.scratch_rows_for(max(64, ref[0].shape[0])), 3), device=gpu)
    before = capi.debug_counters()["streaming_passes"]
    slab = capi.Slab()
    slab.part = 3
    capi.extract_fused_raw(t, thresh, lower, upper, ws, None, None, slab=slab, scratch=scratch)
    slab.part = 4
    capi.extract_fused_raw(t, thresh, lower, upper, ws, None, None, slab=slab, scratch=scratch)
    nv, nf, flags = capi.read_counts(ws, with_flags=True)
    assert (nv, nf) == (ref[0].shape[0], ref[1].shape[0]) and flags == 0
    v = torch.full((nv, 3), float("nan"), device=gpu)
    f = torch.full((nf, 3), -1, dtype=torch.int32, device=gpu)
    if nv:
        slab.part = 6
        capi.extract_fused_raw(t, thresh, lower, upper, ws, v, f if nf else None, slab=slab, scratch=scratch)
    torch.cuda.synchronize()
    assert capi.debug_counters()["streaming_passes"] - before == 1
    keys = vertex_keys_from_workspace(ws.cpu().numpy(), t.shape, nv, capi.debug_layout(*t.shape))
    _assert_same_mesh((v.cpu().numpy(), f.cpu().numpy(), keys), ref)


def test_fused_capacity_overflow_falls_back_to_exact_emit(gpu):
    g, thresh, lower, upper = small_cases()["noise_33x17x200"]
    hip = _hip_extract_fused(gpu, g, thresh, lower, upper, cap_vertices=100, cap_faces=50)
    _assert_same_mesh(hip, oracle_extract(g, thresh, lower, upper))


def test_fused_dense_noise_exercises_direct_path(gpu):
    """White noise makes >1000 vertices per block-plane: LDS stage overflows, waves write directly."""
    g = np.random.default_rng(77).standard_normal((24, 40, 512)).astype(np.float32)
    hip = _hip_extract_fused(gpu, g, 0.0, None, None)
    _assert_same_mesh(hip, oracle_extract(g, 0.0))


def test_fused_wide_rows_need_z_halo(gpu):
    """rz > 512: two z tiles per row, the tile seam goes through the z-halo voxel."""
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid((12, 10, 1100), period=16, seed=9).numpy()
    hip = _hip_extract_fused(gpu, g, 0.0, None, None)
    _assert_same_mesh(hip, oracle_extract(g, 0.0))
    g = np.random.default_rng(5).standard_normal((6, 7, 1030)).astype(np.float32)
    hip = _hip_extract_fused(gpu, g, 0.1, None, None)
    _assert_same_mesh(hip, oracle_extract(g, 0.1))


@pytest.mark.parametrize("shape", [(3, 3, 2100), (4, 2, 16450), (2, 9, 2049)])
def test_long_rows_use_the_wide_face_staging(gpu, shape):
    """rz > 2048 (rows of more than 32 chunks) selects k_faces<256>; rz > 16384 (more than 256 chunks per row) its
    two-range staging; both through the counting call and the one-pass call."""
    g = np.random.default_rng(sum(shape)).standard_normal(shape).astype(np.float32)
    ref = oracle_extract(g, 0.2)
    _assert_same_mesh(_hip_extract(gpu, g, 0.2, None, None), ref)
    _assert_same_mesh(_hip_extract_fused(gpu, g, 0.2, None, None), ref)


def test_fused_fp16(gpu):
    g = small_cases()["perlin48"][0].astype(np.float16)
    hip = _hip_extract_fused(gpu, g, 0.0, None, None, dtype=torch.float16)
    _assert_same_mesh(hip, oracle_extract(g.astype(np.float32), 0.0))


FP16_THRESHOLDS = [0.1, -0.1, 1e-10, -1e-10, 5.9604645e-8, 6e-8, -6e-8, 0.0, -0.0, 65504.0, 65519.9, 65520.0, 70000.0,
                   -65504.0, -65520.0, -70000.0, float("inf"), float("-inf"), float("nan"), 0.333251953125, 1.0009765625]


@pytest.mark.parametrize("thresh", FP16_THRESHOLDS, ids=[repr(t) for t in FP16_THRESHOLDS])
def test_fp16_grid_is_classified_like_its_upcast(gpu, thresh):
    """fp16 grids are compared as 16-bit values against the threshold rounded DOWN to fp16 (p3d_mc.hip,
    half_round_down): that must be exactly the reference's `grid.float() > thresh` (marching_cubes.py:87, cu:25) for every
    threshold -- not representable in fp16, between the subnormals, beyond the fp16 range, infinite, NaN -- and every
    sample value: +-0, subnormals, +-65504, +-inf, NaN.  The whole mesh against the oracle on the up-cast grid."""
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 5.96e-8, -5.96e-8, 6.1e-5, -6.1e-5, 65504.0, -65504.0, np.inf, -np.inf, np.nan, 0.1, -0.1,
                        0.333251953125, 0.33349609375, 1.0, 1.0009765625, 1e-7, -1e-7], dtype=np.float16)
    g = rng.standard_normal((9, 11, 140)).astype(np.float16)
    idx = rng.integers(0, g.size, size=g.size // 3)
    g.reshape(-1)[idx] = special[rng.integers(0, len(special), size=len(idx))]
    ref = oracle_extract(g.astype(np.float32), thresh)
    with np.errstate(invalid="ignore"):
        _assert_same_mesh(_hip_extract_fused(gpu, g, thresh, None, None, dtype=torch.float16), ref)
        _assert_same_mesh(_hip_extract(gpu, g, thresh, None, None, dtype=torch.float16), ref)


def test_fused_medium_perlin_192(gpu):
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid(192).numpy()
    hip = _hip_extract_fused(gpu, g, 0.0, None, None)
    assert hip[0].shape[0] == 268980 and hip[1].shape[0] == 531431
    _assert_same_mesh(hip, oracle_extract(g, 0.0))


def test_pybind_module_matches_oracle(gpu, built):
    g, thresh, lower, upper = small_cases()["noise_33x17x200"]
    t = torch.from_numpy(g).to(gpu)
    v, f = built.libPrim3D.marching_cubes(t, thresh, lower, upper)
    assert v.dtype == torch.float32 and f.dtype == torch.int32 and v.is_cuda and f.is_cuda
    nv, nf = oracle_count(g, thresh)
    assert v.shape == (nv, 3) and f.shape == (nf, 3)
    rv, rf, _ = oracle_extract(g, thresh, lower, upper)
    # without keys: compare the triangle soup (positions per corner), sorted
    soup = lambda vv, ff: np.sort(vv[ff.astype(np.int64)].reshape(len(ff), 9).view([("", np.float32)] * 9), axis=0)
    assert np.array_equal(soup(v.cpu().numpy(), f.cpu().numpy()), soup(rv, rf))


def test_fp16_grid_through_the_wrapper(gpu, built):
    """prim3d.marching_cubes on a float16 grid: the reference's wrapper up-casts it (marching_cubes.py:87) and its C++ entry
    sees float32; here the grid stays float16 all the way into the kernel (half the bytes, no copy).  Same mesh as the
    up-cast -- against the oracle on `grid.float()` and against this module's own float32 call; other dtypes still go
    through float32, and the native module still refuses what the reference's data_ptr<float>() would."""
    g16 = small_cases()["perlin48"][0].astype(np.float16)
    t16 = torch.from_numpy(g16).to(gpu)
    p0 = torch.cuda.memory_allocated(gpu)
    v, f = built.marching_cubes(t16, 0.01)
    v32, f32 = built.marching_cubes(t16.float(), 0.01)
    rv, rf, _ = oracle_extract(g16.astype(np.float32), 0.01)
    soup = lambda vv, ff: np.sort(vv[ff.astype(np.int64)].reshape(len(ff), 9).view([("", np.float32)] * 9), axis=0)
    assert v.dtype == torch.float32 and f.dtype == torch.int32 and v.shape == rv.shape and f.shape == rf.shape
    assert np.array_equal(soup(v.cpu().numpy(), f.cpu().numpy()), soup(rv, rf))
    assert np.array_equal(soup(v32.cpu().numpy(), f32.cpu().numpy()), soup(rv, rf))
    with pytest.raises(RuntimeError, match="expected scalar type Float but found Double"):
        built.libPrim3D.marching_cubes(t16.double(), 0.0, [0.0] * 3, [48.0] * 3)
    vd, fd = built.marching_cubes(t16.double(), 0.01)   # (the wrapper converts every other dtype, as the reference does)
    assert vd.shape == rv.shape and fd.shape == rf.shape
    del p0


def test_reference_example_counts(gpu, built):
    """examples/sphere.py:8-15 through the Python wrapper: int64 grid, thresh 0 -> 11766 / 23528."""
    from primitive3d_amd.fields import sphere_grid
    grid = torch.tensor(sphere_grid(200)).cuda()
    v, f = built.marching_cubes(grid, 0)
    assert v.shape == (11766, 3) and f.shape == (23528, 3)


def test_fp16_grid_equals_upcast(gpu):
    g = small_cases()["perlin48"][0].astype(np.float16)
    hip = _hip_extract(gpu, g, 0.0, None, None, dtype=torch.float16)
    ref = oracle_extract(g.astype(np.float32), 0.0)
    _assert_same_mesh(hip, ref)


def test_empty_result_shapes(gpu, built):
    t = torch.zeros((4, 5, 6), device=gpu)
    v, f = built.libPrim3D.marching_cubes(t, 0.5, [0.0, 0.0, 0.0], [4.0, 5.0, 6.0])
    assert v.shape == (0, 3) and f.shape == (0, 3)


def test_boundary_errors(gpu, built):
    C = built.libPrim3D
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        C.marching_cubes(torch.zeros(4, 4, 4), 0.0, [0.0] * 3, [4.0] * 3)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        C.marching_cubes(torch.zeros(4, 4, 4, device=gpu).permute(2, 1, 0), 0.0, [0.0] * 3, [4.0] * 3)
    with pytest.raises(RuntimeError):
        C.marching_cubes(torch.zeros(4, 4, device=gpu), 0.0, [0.0] * 3, [4.0] * 3)
    with pytest.raises(ValueError):
        built.marching_cubes(torch.zeros(1, 5, 6), 0.0)


def test_medium_perlin_192(gpu):
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid(192).numpy()
    hip = _hip_extract(gpu, g, 0.0, None, None)
    ref = oracle_extract(g, 0.0)
    assert hip[0].shape[0] == 268980 and hip[1].shape[0] == 531431
    _assert_same_mesh(hip, ref)


def test_counts_without_the_mailbox(gpu):
    """P3D_NO_MAILBOX=1 (read once per process): p3d_mc_read_counts falls back to copy + synchronise."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r)\n"
            "from primitive3d_amd import capi\n"
            "from tests.cases import small_cases\n"
            "g, t, lo, up = small_cases()['noise_33x17x200']\n"
            "v, f = capi.extract_fused(torch.from_numpy(g).cuda(), t, lo, up)\n"
            "v2, f2 = capi.extract(torch.from_numpy(g).cuda(), t, lo, up)[:2]\n"
            "print(v.shape[0], f.shape[0], v2.shape[0], f2.shape[0])\n") % str(root)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "P3D_NO_MAILBOX": "1"}, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    g, t, _, _ = small_cases()["noise_33x17x200"]
    nv, nf = oracle_count(g, t)
    assert out.stdout.split() == [str(nv), str(nf), str(nv), str(nf)]


@pytest.mark.parametrize("early", ["0", "1", "7"])
def test_vertex_copy_split_between_the_two_face_launches(gpu, tuning_env, early):
    """P3D_COMPACT_EARLY slices of every vertex region are copied by blocks riding with the counting kernel, the
    rest by blocks riding with k_faces: any split gives the same mesh."""
    tuning_env("P3D_COMPACT_EARLY", early)
    g, thresh, lower, upper = small_cases()["perlin_40x24x96_thr"]
    _assert_same_mesh(_hip_extract_fused(gpu, g, thresh, lower, upper), oracle_extract(g, thresh, lower, upper))


def test_adapter_on_a_row_with_a_nearly_empty_last_z_tile(gpu, built):
    """rz = 517: rows of 8 full chunks + 5 voxels, i.e. a second z tile of the streaming kernel that carries almost
    nothing.  The vertices then spread unevenly over the scratch regions; the adapter must still return the exact
    mesh (growing its per-region headroom), call after call."""
    g = np.random.default_rng(21).standard_normal((24, 40, 517)).astype(np.float32)
    nv, nf = oracle_count(g, 0.1)
    t = torch.from_numpy(g).to(gpu)
    for _ in range(3):
        v, f = built.libPrim3D.marching_cubes(t, 0.1, [0.0, 0.0, 0.0], [24.0, 40.0, 517.0])
        assert v.shape == (nv, 3) and f.shape == (nf, 3)
    rv, rf, _ = oracle_extract(g, 0.1, [0.0, 0.0, 0.0], [24.0, 40.0, 517.0])
    soup = lambda vv, ff: np.sort(vv[ff.astype(np.int64)].reshape(len(ff), 9).view([("", np.float32)] * 9), axis=0)
    assert np.array_equal(soup(v.cpu().numpy(), f.cpu().numpy()), soup(rv, rf))


@pytest.mark.dev_hooks
def test_region_id_space_overflow_falls_back_to_the_counting_call(gpu, built, tuning_env):
    """A region that numbers more than 2^26 vertices makes the one-pass ids ambiguous (include/p3d_mc.h,
    p3d_mc_read_counts bit 1).  P3D_TEST_ID_LIMIT pretends the id space is tiny: the flag must come back, and both the
    ctypes flow and the pybind adapter must renumber with p3d_mc_count + p3d_mc_emit and still return the exact mesh."""
    from primitive3d_amd import capi
    tuning_env("P3D_TEST_ID_LIMIT", "16")
    g, thresh, lower, upper = small_cases()["noise_33x17x200"]
    t = torch.from_numpy(g).to(gpu)
    ws = torch.empty(capi.workspace_bytes(*t.shape), dtype=torch.uint8, device=gpu)
    capi.extract_fused_raw(t, thresh, lower, upper, ws, None, None)
    nv, nf, flags = capi.read_counts(ws, with_flags=True)
    assert (nv, nf) == oracle_count(g, thresh) and flags & 2
    ref = oracle_extract(g, thresh, lower, upper)
    v, f = capi.extract_fused(t, thresh, lower, upper)
    soup = lambda vv, ff: np.sort(vv[ff.astype(np.int64)].reshape(len(ff), 9).view([("", np.float32)] * 9), axis=0)
    assert np.array_equal(soup(v.cpu().numpy(), f.cpu().numpy()), soup(ref[0], ref[1]))
    v, f = built.libPrim3D.marching_cubes(t, thresh, lower, upper)
    assert np.array_equal(soup(v.cpu().numpy(), f.cpu().numpy()), soup(ref[0], ref[1]))
    tuning_env("P3D_TEST_ID_LIMIT", None)
    capi.extract_fused_raw(t, thresh, lower, upper, ws, None, None)
    assert capi.read_counts(ws, with_flags=True)[2] == 0


def test_ply_of_a_device_resident_extraction(gpu, built, tmp_path):
    """save_mesh on the CUDA tensors marching_cubes returns (examples/sphere.py:15-17) == the restated reference
    writer (oracle/ply_oracle.py, marching_cubes.cu:307-352) byte for byte."""
    from oracle.ply_oracle import reference_ply_bytes
    from primitive3d_amd.fields import sphere_grid
    v, f = built.marching_cubes(sphere_grid(64), 0, scale=2.0)
    assert v.is_cuda and f.is_cuda
    p = tmp_path / "sphere.ply"
    built.save_mesh(v, f, filename=p)
    vn, fn = v.cpu().numpy(), f.cpu().numpy()
    assert p.read_bytes() == reference_ply_bytes(vn, fn, np.full(vn.shape, 127, np.uint8))
    colors = (torch.rand(v.shape, device=gpu) * 255).to(torch.uint8)
    built.libPrim3D.save_mesh_as_ply(str(p), v, f, colors)
    assert p.read_bytes() == reference_ply_bytes(vn, fn, colors.cpu().numpy())


@pytest.mark.parametrize("env", [{"P3D_MC_MODE": "exact"}, {"P3D_MC_MODE": "hinted"}, {"P3D_MC_MODE": "scratch"}, {},
                                 {"P3D_NO_MAILBOX": "1"},   # (the totals and the region totals by copy + synchronise)
                                 {"P3D_MC_MODE": "fast"}])
def test_adapter_modes_in_a_fresh_process(gpu, env):
    """The pybind adapter's one switch is read once per process: `P3D_MC_MODE=exact` (the reference's own order: count,
    read, allocate exactly, emit), `hinted` (the default: from the third call on a field that stands still, the vertices are
    stored where they stay) or `scratch` (hinted, always through the scratch tensor).  Each must give the oracle's counts on
    repeated calls; with exact allocations the storage sizes are exact; any other value is an error, not a silent default."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "import primitive3d_amd as p3d\n"
            "from tests.cases import small_cases\n"
            "g, t, lo, up = small_cases()['noise_33x17x200']\n"
            "x = torch.from_numpy(g).cuda()\n"
            "for _ in range(5):\n"
            "    v, f = p3d.libPrim3D.marching_cubes(x, t, lo, up)\n"
            "torch.cuda.synchronize()\n"
            "print(v.shape[0], f.shape[0], v.untyped_storage().nbytes() // 12, f.untyped_storage().nbytes() // 12)\n") % str(root)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    if env.get("P3D_MC_MODE") == "fast":
        assert out.returncode != 0 and "P3D_MC_MODE must be 'hinted', 'scratch' or 'exact'" in out.stderr, out.stderr[-2000:]
        return
    assert out.returncode == 0, out.stderr[-2000:]
    g, t, _, _ = small_cases()["noise_33x17x200"]
    nv, nf = oracle_count(g, t)
    got = [int(s) for s in out.stdout.split()]
    assert got[:2] == [nv, nf]
    if env.get("P3D_MC_MODE") == "exact":   # exact allocations: the storage holds exactly the rows
        assert got[2:] == [nv, nf], got
    else:     # default: rows [0, V) of a buffer that may be up to 1/8 + 4096 rows longer
        assert nv <= got[2] <= nv + nv // 8 + 4096 and nf <= got[3] <= nf + nf // 8 + 4096, got


@pytest.mark.parametrize("knob", ["0", "1"])
def test_face_tiles_scalar_precheck_both_ways(gpu, knob):
    """k_faces asks for its tile's triangle count by a scalar load first when the launch expects few triangles per tile
    (FaceArgs::sparse; by default decided from the face capacity, i.e. after the first call on a sphere-in-a-box field).
    P3D_FACES_SPARSE forces it off / on for every launch (read once per process): a sparse and a dense field must give the
    same meshes either way -- compared with the oracle's as sorted triangle soups."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = ("import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
            "import primitive3d_amd as p3d\n"
            "from bench import soup_hashes\n"
            "from primitive3d_amd.fields import sphere_grid, perlin_grid\n"
            "from oracle import oracle_extract\n"
            "for name, g in (('sphere', torch.tensor(sphere_grid(96)).float().cuda()), ('noise', perlin_grid((70, 64, 130), period=16, seed=5, device='cuda'))):\n"
            "    up = [float(s) for s in g.shape]\n"
            "    for _ in range(3):\n"
            "        v, f = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)\n"
            "    ov, of = oracle_extract(g.cpu().numpy(), 0.0, [0.0] * 3, up)[:2]\n"
            "    a = soup_hashes(v, f); b = soup_hashes(torch.from_numpy(ov).cuda(), torch.from_numpy(of.astype(np.int32)).cuda())\n"
            "    assert v.shape[0] == ov.shape[0] and f.shape[0] == of.shape[0] and f.shape[0] > 0, name\n"
            "    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), name\n"
            "print('ok')\n") % str(root)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "P3D_FACES_SPARSE": knob}, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_threads_sharing_one_stream_through_the_c_abi(gpu, built):
    """Four host threads enqueue whole extractions on ONE stream through ctypes (which releases the GIL, so the calls
    really interleave on the host): the library hands every call a block of the stream's cursor ring and the streaming
    kernel of the call before it clears that block, so the launches of a call must be enqueued as a whole, in ring
    order (per-stream lock, include/p3d_mc.h "Threads").  Every result must equal the single-threaded one."""
    import threading
    from bench import soup_hashes
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    shapes = [(96, 80, 130), (64, 64, 64), (128, 128, 200), (40, 200, 520)]
    grids = [perlin_grid(s, period=24, seed=10 + i, device=gpu) for i, s in enumerate(shapes)]
    ref = []
    for g in grids:
        v, f = capi.extract_fused(g, 0.0)
        torch.cuda.synchronize()
        ref.append((v.shape[0], f.shape[0], soup_hashes(v, f)[0]))
    st = torch.cuda.Stream(device=gpu)
    errors, results = [], []
    lock = threading.Lock()

    def worker(tid):
        try:
            with torch.cuda.stream(st):   # (the current stream is per thread: every thread selects the shared one)
                for it in range(30):
                    k = (it * 3 + tid) % len(grids)
                    v, f = capi.extract_fused(grids[k], 0.0)
                    if (v.shape[0], f.shape[0]) != ref[k][:2]:
                        errors.append((tid, it, k, tuple(v.shape), tuple(f.shape)))
                    elif it % 5 == 0:
                        with lock:
                            results.append((k, v, f))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    st.synchronize()
    torch.cuda.synchronize()
    assert not errors, errors[:5]
    for k, v, f in results:
        assert torch.equal(soup_hashes(v, f)[0], ref[k][2]), f"grid {k}: triangle soup differs"


def test_threads_on_their_own_streams(gpu, built):
    """Three host threads, each on its own stream, call the adapter concurrently on different grids (per-stream cursor
    rings, per-call mailbox slots, the shape-hint map): every result equals the single-threaded result of its grid."""
    import threading
    from bench import soup_hashes
    from primitive3d_amd.fields import perlin_grid
    shapes = [(96, 80, 130), (64, 64, 64), (128, 128, 200), (40, 200, 520)]
    grids = [perlin_grid(s, period=24, seed=i, device=gpu) for i, s in enumerate(shapes)]
    ref = []
    for g in grids:
        v, f = built.marching_cubes(g, 0.0)
        torch.cuda.synchronize()
        ref.append((v.shape[0], f.shape[0], soup_hashes(v, f)[0]))
    errors = []

    def worker(tid):
        st = torch.cuda.Stream(device=gpu)
        with torch.cuda.stream(st):
            for it in range(40):
                k = (it * 3 + tid) % len(grids)
                v, f = built.marching_cubes(grids[k], 0.0)
                if (v.shape[0], f.shape[0]) != ref[k][:2]:
                    errors.append((tid, it, k, tuple(v.shape), tuple(f.shape)))
                elif it % 8 == 0:
                    st.synchronize()
                    if not torch.equal(soup_hashes(v, f)[0], ref[k][2]):
                        errors.append((tid, it, k, "triangle soup differs"))
        st.synchronize()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors[:5]
