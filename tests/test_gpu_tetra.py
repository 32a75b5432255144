"""GPU parity of the HIP marching tetrahedra (libp3dmt.so, through the reference-shaped wrapper) against outputs of
the REFERENCE itself (tests/golden/tetra_*.npz) and against the oracle restatement on larger seeded meshes."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle.mt_oracle import mt_oracle

pytestmark = pytest.mark.gpu
GOLD = sorted((Path(__file__).parent / "golden").glob("tetra_*.npz"))
TOL = 1e-6  # positions are asserted bit-identical; TOL only words the failure message


def _same(v, f, ti, ta, ref_v, ref_f, ref_ti, ref_ta):
    assert np.array_equal(ta, ref_ta), "orientation fix differs"
    assert np.array_equal(f, ref_f), "faces differ"
    assert np.array_equal(ti, ref_ti), "tet indices differ"
    assert v.shape == ref_v.shape
    if not np.array_equal(v, ref_v):
        d = np.abs(v.astype(np.float64) - ref_v).max()
        pytest.fail(f"vertex positions differ by {d} (tolerance for information: {TOL})")


@pytest.mark.parametrize("path", GOLD, ids=[p.stem for p in GOLD])
@pytest.mark.parametrize("where", ["cuda", "cpu"])
def test_reference_vectors(gpu, built, path, where):
    g = np.load(path)
    dev = gpu if where == "cuda" else torch.device("cpu")
    tets = torch.from_numpy(g["tets"].copy()).to(dev)
    v, f, ti = built.marching_tetrahedras(torch.from_numpy(g["points"]).to(dev), tets, torch.from_numpy(g["sdf"]).to(dev), True)
    assert v.device.type == where and f.dtype == torch.int64 and v.dtype == torch.float32
    _same(v.cpu().numpy(), f.cpu().numpy(), ti.cpu().numpy(), tets.cpu().numpy(), g["verts"], g["faces"], g["tet_idx"],
          g["tets_after"])
    v2, f2 = built.marching_tetrahedras(torch.from_numpy(g["points"]).to(dev), tets, torch.from_numpy(g["sdf"]).to(dev))
    assert torch.equal(v2, v) and torch.equal(f2, f)   # (tets are already oriented now: idempotent)


SLIVERS = sorted((Path(__file__).parent / "golden").glob("tetraslivers_*.npz"))


@pytest.mark.parametrize("path", SLIVERS, ids=[p.stem for p in SLIVERS])
def test_reference_vectors_with_slivers(gpu, built, path):
    """The reference's outputs on meshes that keep their slivers: the library (float64 determinant) may orient only
    tets whose determinant is rounding noise differently from the reference's float32 LU (tests/tetra_compare.py); the
    same bar against the oracle."""
    from tests.tetra_compare import assert_equal_modulo_flat_tets
    g = np.load(path)
    tets = torch.from_numpy(g["tets"].copy()).to(gpu)
    v, f, ti = built.marching_tetrahedras(torch.from_numpy(g["points"]).to(gpu), tets, torch.from_numpy(g["sdf"]).to(gpu), True)
    out = (v.cpu().numpy(), f.cpu().numpy(), ti.cpu().numpy(), tets.cpu().numpy())
    assert_equal_modulo_flat_tets(g["points"], g["tets"], out, (g["verts"], g["faces"], g["tet_idx"], g["tets_after"]),
                                  max_differing=8)
    # (library and oracle both take float64 determinants, but by different formulas -- cofactor expansion vs LAPACK's
    #  LU: on rounding-noise determinants they too may differ, and only there)
    assert_equal_modulo_flat_tets(g["points"], g["tets"], out, mt_oracle(g["points"], g["tets"], g["sdf"]), max_differing=16)


def _grid_tets(n, seed):
    """n^3 cubes of 5 tetrahedra each on a jittered lattice (an SDF-friendly mesh with hundreds of thousands of tets)."""
    rng = np.random.default_rng(seed)
    ax = np.arange(n + 1)
    P = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    P += rng.uniform(-0.2, 0.2, P.shape).astype(np.float32)
    idx = lambda x, y, z: (x * (n + 1) + y) * (n + 1) + z
    c = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    v = [idx(x + dx, y + dy, z + dz) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)]   # v[4dx+2dy+dz]
    par = ((x + y + z) % 2 == 0)
    A = [(0, 3, 5, 6), (0, 1, 3, 5), (0, 2, 3, 6), (0, 4, 5, 6), (3, 5, 6, 7)]
    B = [(1, 2, 4, 7), (0, 1, 2, 4), (1, 2, 3, 7), (1, 4, 5, 7), (2, 4, 6, 7)]
    tets = []
    for ta, tb in zip(A, B):
        tets.append(np.where(par[:, None], np.stack([v[k] for k in ta], 1), np.stack([v[k] for k in tb], 1)))
    T = np.concatenate(tets).astype(np.int64)
    T = T[rng.permutation(len(T))]
    sdf = (np.linalg.norm(P - n / 2, axis=1) - n / 3 + 0.3 * np.sin(P[:, 0])).astype(np.float32)
    return P, T, sdf


@pytest.mark.parametrize("n", [12, 40])
def test_lattice_meshes_match_the_oracle(gpu, built, n):
    P, T, sdf = _grid_tets(n, n)
    tets = torch.from_numpy(T.copy()).to(gpu)
    v, f, ti = built.marching_tetrahedras(torch.from_numpy(P).to(gpu), tets, torch.from_numpy(sdf).to(gpu), True)
    rv, rf, rti, rta = mt_oracle(P, T, sdf)
    _same(v.cpu().numpy(), f.cpu().numpy(), ti.cpu().numpy(), tets.cpu().numpy(), rv, rf, rti, rta)
    assert f.numel() == 0 or (int(f.min()) >= 0 and int(f.max()) < v.shape[0])


def test_gradients_flow_like_the_reference(gpu, built):
    """vertices and sdf get gradients through the interpolation (:178-190); checked against autograd on the oracle's
    formula re-done in float64."""
    g = np.load(Path(__file__).parent / "golden" / "tetra_delaunay_smooth_400.npz")
    P = torch.from_numpy(g["points"]).to(gpu).requires_grad_(True)
    S = torch.from_numpy(g["sdf"]).to(gpu).requires_grad_(True)
    tets = torch.from_numpy(g["tets"].copy()).to(gpu)
    v, f = built.marching_tetrahedras(P, tets, S)
    assert v.requires_grad and np.array_equal(f.cpu().numpy(), g["faces"])
    assert torch.allclose(v.detach().cpu(), torch.from_numpy(g["verts"]), atol=1e-6)
    (v ** 2).sum().backward()
    assert P.grad is not None and S.grad is not None and torch.isfinite(P.grad).all() and torch.isfinite(S.grad).all()
    assert float(P.grad.abs().sum()) > 0 and float(S.grad.abs().sum()) > 0


def test_argument_errors(gpu, built):
    P = torch.zeros(4, 3, device=gpu)
    T = torch.tensor([[0, 1, 2, 3]], device=gpu)
    S = torch.zeros(4, device=gpu)
    with pytest.raises(TypeError):
        built.marching_tetrahedras(P.double(), T, S)
    with pytest.raises(TypeError):
        built.marching_tetrahedras(P, T.int(), S)
    with pytest.raises(ValueError):
        built.marching_tetrahedras(P, T[:, :3], S)
    v, f = built.marching_tetrahedras(P, T, S)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    # a tet index outside [0, N): the reference's `vertices[tets]` raises IndexError; here it is never dereferenced
    S2 = torch.tensor([-1.0, 1.0, 1.0, 1.0], device=gpu)
    for bad in ([[0, 1, 2, 4]], [[0, -1, 2, 3]], [[0, 1, 2, 3], [1 << 40, 1, 2, 3]]):
        with pytest.raises(IndexError):
            built.marching_tetrahedras(P, torch.tensor(bad, device=gpu), S2)
    v, f = built.marching_tetrahedras(torch.rand(4, 3, device=gpu), T.clone(), S2)   # (the library still works afterwards)
    assert v.shape == (3, 3) and f.shape == (1, 3)


@pytest.mark.parametrize("script,args", [("sphere.py", ["--out", ""]), ("bunny_sdf.py", ["--out", ""]),
                                         ("sphere_tetrahedra.py", [""])])
def test_example_scripts_run(gpu, script, args):
    """The acceptance scripts of the reference (examples/*.py), ported: each asserts the reference's recorded counts."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(root / "examples" / script), *args], capture_output=True, text=True,
                         timeout=600, cwd=str(root))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]


def test_unaligned_tets_and_repeated_emit(gpu, built):
    """A tet array that starts 8 bytes off a 16-byte boundary takes the scalar-load classify kernel; p3d_mt_emit called a
    second time (without the host-side copy of the sizes, which the first call consumed) reads them from the workspace."""
    import ctypes
    from primitive3d_amd import tetrahedra as T
    P, tets_np, sdf = _grid_tets(10, 3)
    flat = torch.zeros(tets_np.size + 1, dtype=torch.int64, device=gpu)
    tets = flat[1:].view(-1, 4)
    tets.copy_(torch.from_numpy(tets_np))
    assert tets.data_ptr() % 16 == 8
    pts, s = torch.from_numpy(P).to(gpu), torch.from_numpy(sdf).to(gpu)
    v, f, ti = built.marching_tetrahedras(pts, tets, s, True)
    rv, rf, rti, rta = mt_oracle(P, tets_np, sdf)
    _same(v.cpu().numpy(), f.cpu().numpy(), ti.cpu().numpy(), tets.cpu().numpy(), rv, rf, rti, rta)
    # the C ABI directly: prepare once, emit twice
    L = T.lib()
    nb = ctypes.c_size_t(0)
    assert L.p3d_mt_workspace_bytes(pts.shape[0], tets.shape[0], ctypes.byref(nb)) == 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device=gpu)
    nv, nf = ctypes.c_int64(0), ctypes.c_int64(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.p3d_mt_prepare(p(pts), pts.shape[0], p(tets), tets.shape[0], p(s), p(ws), ctypes.byref(nv), ctypes.byref(nf), st) == 0
    assert (nv.value, nf.value) == (rv.shape[0], rf.shape[0])
    for _ in range(2):
        ov = torch.empty((nv.value, 3), device=gpu)
        of = torch.empty((nf.value, 3), dtype=torch.int64, device=gpu)
        assert L.p3d_mt_emit(p(pts), p(tets), p(s), p(ws), p(ov), None, p(of), None, st) == 0
        torch.cuda.synchronize()
        assert np.array_equal(ov.cpu().numpy(), rv) and np.array_equal(of.cpu().numpy(), rf)


def test_many_tets_around_one_edge(gpu, built):
    """A fan of K tetrahedra around one crossing edge (every tet inserts the same key into the hash set) plus a ring of
    crossing edges each shared by two tets."""
    K = 300
    ang = np.linspace(0, 2 * np.pi, K, endpoint=False)
    ring = np.stack([np.cos(ang), np.sin(ang), np.zeros(K)], 1)
    P = np.concatenate([[[0, 0, -1.0], [0, 0, 1.0]], ring]).astype(np.float32)
    T = np.stack([np.zeros(K), np.ones(K), 2 + np.arange(K), 2 + (np.arange(K) + 1) % K], 1).astype(np.int64)
    sdf = np.concatenate([[-1.0, 0.7], 0.3 + 0.2 * np.cos(3 * ang)]).astype(np.float32)
    tets = torch.from_numpy(T.copy()).to(gpu)
    v, f, ti = built.marching_tetrahedras(torch.from_numpy(P).to(gpu), tets, torch.from_numpy(sdf).to(gpu), True)
    rv, rf, rti, rta = mt_oracle(P, T, sdf)
    assert rv.shape[0] == K + 1
    _same(v.cpu().numpy(), f.cpu().numpy(), ti.cpu().numpy(), tets.cpu().numpy(), rv, rf, rti, rta)


def test_special_sdf_values(gpu, built):
    """Exact zeros (outside: sdf > 0 is false, :151), infinities, NaNs and subnormals in the SDF: same topology as the
    oracle, positions equal with NaN == NaN (a NaN's payload is not specified)."""
    P, T, sdf = _grid_tets(9, 21)
    rng = np.random.default_rng(5)
    pick = rng.permutation(len(sdf))
    sdf[pick[:60]] = 0.0
    sdf[pick[60:90]] = np.inf
    sdf[pick[90:120]] = -np.inf
    sdf[pick[120:150]] = np.nan
    sdf[pick[150:220]] *= np.float32(1e-42)
    tets = torch.from_numpy(T.copy()).to(gpu)
    v, f, ti = built.marching_tetrahedras(torch.from_numpy(P).to(gpu), tets, torch.from_numpy(sdf).to(gpu), True)
    with np.errstate(all="ignore"):
        rv, rf, rti, rta = mt_oracle(P, T, sdf)
    assert np.array_equal(tets.cpu().numpy(), rta) and np.array_equal(f.cpu().numpy(), rf)
    assert np.array_equal(ti.cpu().numpy(), rti)
    assert np.array_equal(v.cpu().numpy(), rv, equal_nan=True)
    assert np.isnan(rv).any() and np.isfinite(rv).any()


def test_threads_on_their_own_streams(gpu, built):
    """Concurrent calls from three host threads, each on its own stream (pinned size slots, the host-side copy of the
    sizes between the two phases): every result equals the single-threaded one."""
    import threading
    meshes = []
    for n, seed in ((8, 1), (12, 2), (15, 3)):
        P, T, sdf = _grid_tets(n, seed)
        tets = torch.from_numpy(T.copy()).to(gpu)
        args = (torch.from_numpy(P).to(gpu), tets, torch.from_numpy(sdf).to(gpu))
        v, f = built.marching_tetrahedras(*args)   # (also orients the tets once)
        torch.cuda.synchronize()
        meshes.append((args, v.clone(), f.clone()))
    errors = []

    def worker(tid):
        st = torch.cuda.Stream(device=gpu)
        with torch.cuda.stream(st):
            for it in range(30):
                args, rv, rf = meshes[(it + tid) % len(meshes)]
                v, f = built.marching_tetrahedras(*args)
                st.synchronize()
                if not (torch.equal(v, rv) and torch.equal(f, rf)):
                    errors.append((tid, it))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
