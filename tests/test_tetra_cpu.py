"""The marching-tetrahedra oracle (oracle/mt_oracle.py) against outputs of the REFERENCE itself
(tests/golden/tetra_*.npz, tools/gen_tetra_goldens.py): this oracle is pinned by reference-run vectors."""
from pathlib import Path

import numpy as np
import pytest

from oracle.mt_oracle import mt_oracle

GOLD = sorted((Path(__file__).parent / "golden").glob("tetra_*.npz"))


@pytest.mark.parametrize("path", GOLD, ids=[p.stem for p in GOLD])
def test_oracle_reproduces_reference_outputs(path):
    g = np.load(path)
    v, f, ti, ta = mt_oracle(g["points"], g["tets"], g["sdf"])
    assert np.array_equal(ta, g["tets_after"]), "orientation fix differs"
    assert np.array_equal(f, g["faces"]) and np.array_equal(ti, g["tet_idx"])
    assert v.shape == g["verts"].shape
    assert np.array_equal(v, g["verts"]), np.abs(v - g["verts"]).max()


def test_goldens_cover_the_reference_example():
    g = np.load(Path(__file__).parent / "golden" / "tetra_example_sphere.npz")
    assert g["points"].shape == (2056, 3) and g["tets"].shape == (12045, 4) and g["verts"].shape[0] == 4650


@pytest.mark.parametrize("path", GOLD, ids=[p.stem for p in GOLD])
def test_tensor_op_baseline_reproduces_reference_outputs(path):
    """oracle/mt_torch.py (the op chain tools/bench_next.py times beside the HIP library) is pinned the same way."""
    import torch
    from oracle.mt_torch import mt_torch
    g = np.load(path)
    tets = torch.from_numpy(g["tets"].copy())
    v, f, ti = mt_torch(torch.from_numpy(g["points"]), tets, torch.from_numpy(g["sdf"]), True)
    assert np.array_equal(tets.numpy(), g["tets_after"])
    assert np.array_equal(f.numpy(), g["faces"]) and np.array_equal(ti.numpy(), g["tet_idx"])
    assert np.array_equal(v.numpy(), g["verts"])
