"""Comparison of two marching-tetrahedra results on meshes that keep their slivers (tests/golden/tetraslivers_*.npz).

The reference orients a tet by the sign of a float32 `torch.det` of [1, x, y, z] (prim3d/utility/marching_tetrahedras.py:
50-65, an LU factorisation); the oracle and the HIP library take a float64 determinant of the float32 coordinates
(oracle/mt_oracle.py, csrc/p3d_mt.hip k_mt_classify).  On a tet whose true volume is rounding noise the two can disagree
in sign: such a tet then keeps / swaps its first two corners differently, and its triangles come out with the opposite
winding.  Nothing else may differ: same vertices, same face order, same tet indices."""
import numpy as np

FLAT_DET = 1e-12   # |det[p1-p0; p2-p0; p3-p0]| (float64, unit-sized meshes) below which a tet counts as flat


def assert_equal_modulo_flat_tets(points, tets, out, ref, max_differing):
    """out / ref = (verts, faces, tet_idx, tets_after).  Returns the number of tets oriented differently."""
    v, f, ti, ta = out
    rv, rf, rti, rta = ref
    p = points[tets].astype(np.float64)
    det = np.linalg.det(p[:, 1:] - p[:, :1])
    differ = (ta != rta).any(1)
    n = int(differ.sum())
    assert n <= max_differing, f"{n} tets oriented differently from the reference"
    assert n == 0 or float(np.abs(det[differ]).max()) < FLAT_DET, "a well-shaped tet is oriented differently"
    # a differing tet has exactly its first two corners exchanged
    assert np.array_equal(ta[differ][:, [1, 0, 2, 3]], rta[differ])
    assert np.array_equal(v, rv), "vertices do not depend on the orientation: they must be identical"
    assert np.array_equal(ti, rti), "face order / tet indices must be identical"
    same = ~differ[ti]
    assert np.array_equal(f[same], rf[same])
    assert np.array_equal(np.sort(f[~same], axis=1), np.sort(rf[~same], axis=1)), "a flipped tet keeps its triangle's vertices"
    return n
