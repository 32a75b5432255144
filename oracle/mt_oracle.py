"""numpy restatement of the reference's marching tetrahedra, prim3d/utility/marching_tetrahedras.py:89-235.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED: tests/test_tetra_cpu.py checks this file against outputs of the reference itself (tests/golden/tetra_*.npz,
made by tools/gen_tetra_goldens.py, which imports the reference's pure-PyTorch module in the build container).

Steps, with the reference lines they follow:
  :147-148  orientation fix: tets whose [1,x,y,z] determinant is negative get their first two corners swapped IN PLACE
  :151-154  occupancy = sdf > 0; a tet is valid unless its four corners agree
  :157-160  the six edges (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) of every valid tet, each sorted, then unique rows
            (lexicographic order) with the inverse map
  :163-171  vertex ids: rank of a unique edge among those whose endpoints differ in occupancy, -1 for the others
  :178-190  vertex = (v_a * (-s_b) + v_b * s_a) / (s_a - s_b), in float32: weights first, then two products and a sum
  :194-224  faces by the 16-case table; all one-triangle tets first (in tet order), then the two-triangle tets
  :226-234  tet index of every face
"""
import numpy as np

TRIANGLE_TABLE = np.array([  # :8-29
    [-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4],
    [3, 1, 5, -1, -1, -1], [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1],
    [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1],
    [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]], dtype=np.int64)
NUM_TRIANGLES = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0], dtype=np.int64)   # :31-34
BASE_TET_EDGES = np.array([0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3], dtype=np.int64)              # :35-45


def mt_oracle(vertices: np.ndarray, tets: np.ndarray, sdf: np.ndarray):
    """Returns (verts f32 [V,3], faces i64 [F,3], tet_idx i64 [F], tets_after i64 [T,4])."""
    vertices = np.asarray(vertices, np.float32)
    sdf = np.asarray(sdf, np.float32)
    tets = np.array(tets, dtype=np.int64, copy=True)
    # :147-148 (the reference takes a float32 LU determinant; float64 here -- the sign agrees on non-degenerate cells)
    m = np.concatenate([np.ones(tets.shape + (1,)), vertices[tets].astype(np.float64)], axis=-1)
    flip = np.linalg.det(m) < 0
    tets[flip, :2] = tets[flip][:, [1, 0]]
    occ = sdf > 0
    occ4 = occ[tets]
    s = occ4.sum(-1)
    valid = (s > 0) & (s < 4)
    edges = tets[valid][:, BASE_TET_EDGES].reshape(-1, 2)
    edges = np.stack([edges.min(1), edges.max(1)], axis=1)                       # :67-83
    if len(edges):
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
    else:
        uniq, inv = np.zeros((0, 2), np.int64), np.zeros((0,), np.int64)
    cross = occ[uniq].sum(-1) == 1 if len(uniq) else np.zeros((0,), bool)
    mapping = np.full(len(uniq), -1, np.int64)
    mapping[cross] = np.arange(int(cross.sum()))
    idx_map = mapping[inv].reshape(-1, 6)
    pairs = uniq[cross]
    p = vertices[pairs]                                                          # [V,2,3]
    sd = sdf[pairs].copy()                                                       # [V,2]
    sd[:, 1] *= np.float32(-1)
    den = (sd[:, 0] + sd[:, 1])[:, None]
    w = sd[:, ::-1] / den
    verts = (p[:, 0] * w[:, 0:1] + p[:, 1] * w[:, 1:2]).astype(np.float32)
    case = (occ4[valid] * (2 ** np.arange(4))).sum(-1)
    ntri = NUM_TRIANGLES[case]
    one, two = ntri == 1, ntri == 2
    f1 = np.take_along_axis(idx_map[one], TRIANGLE_TABLE[case[one]][:, :3], axis=1) if one.any() else np.zeros((0, 3), np.int64)
    f2 = (np.take_along_axis(idx_map[two], TRIANGLE_TABLE[case[two]], axis=1).reshape(-1, 3)
          if two.any() else np.zeros((0, 3), np.int64))
    faces = np.concatenate([f1.reshape(-1, 3), f2], axis=0)
    tid = np.arange(len(tets))[valid]
    tet_idx = np.concatenate([tid[one], np.repeat(tid[two], 2)])
    return verts.reshape(-1, 3), faces, tet_idx, tets
