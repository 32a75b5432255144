"""dev: randomized parity fuzz of the HIP marching tetrahedra against the oracle (random Delaunay meshes, random SDFs with
exact zeros, shuffled / duplicated tets)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from scipy.spatial import Delaunay
import primitive3d_amd as p3d
from oracle.mt_oracle import mt_oracle
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
gpu = torch.device("cuda", 0)
n_ok = 0
for it in range(int(os.environ.get("N", "40"))):
    npts = int(rng.integers(5, 3000))
    P = rng.standard_normal((npts, 3)).astype(np.float32)
    T = Delaunay(P.astype(np.float64)).simplices.astype(np.int64)
    if rng.random() < 0.3:   # duplicated and shuffled tets
        T = np.concatenate([T, T[rng.integers(0, len(T), len(T) // 3)]])
    T = T[rng.permutation(len(T))]
    kind = rng.integers(0, 3)
    if kind == 0:
        sdf = (np.linalg.norm(P, axis=1) - rng.uniform(0.3, 1.5)).astype(np.float32)
    elif kind == 1:
        sdf = rng.standard_normal(npts).astype(np.float32)
    else:
        sdf = rng.integers(-1, 2, npts).astype(np.float32)   # many exact zeros
    tets = torch.from_numpy(T.copy()).to(gpu)
    v, f, ti = p3d.marching_tetrahedras(torch.from_numpy(P).to(gpu), tets, torch.from_numpy(sdf).to(gpu), True)
    rv, rf, rti, rta = mt_oracle(P, T, sdf)
    assert np.array_equal(tets.cpu().numpy(), rta), (it, "orientation")
    assert np.array_equal(f.cpu().numpy(), rf) and np.array_equal(ti.cpu().numpy(), rti), (it, "faces")
    assert np.array_equal(v.cpu().numpy(), rv, equal_nan=True), (it, "positions")
    n_ok += 1
print("tetra fuzz ok:", n_ok, "cases")
