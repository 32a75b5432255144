"""The reference's marching-tetrahedra demo (examples/sphere_tetrahedra.py) on the MI355X build: the same data
(points / sdfs / tetrahedra of a sphere, kept as inputs in tests/golden/tetra_example_sphere.npz), the same two runs
(tensors on the host, tensors on the GPU), counts checked against what the reference itself produced."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import prim3d  # noqa: E402

if __name__ == "__main__":
    data = np.load(REPO / "tests" / "golden" / "tetra_example_sphere.npz")
    points, sdfs = torch.from_numpy(data["points"]), torch.from_numpy(data["sdf"])
    tets = torch.from_numpy(data["tets"].copy()).long()
    with prim3d.Timer("host tensors (moved to the GPU and back): {:.6f}s"):
        verts, faces = prim3d.marching_tetrahedras(points, tets, sdfs)
    points, sdfs, tets = points.cuda(), sdfs.cuda(), tets.cuda()
    for _ in range(2):
        with prim3d.Timer("device tensors: {:.6f}s"):
            verts, faces = prim3d.marching_tetrahedras(points, tets, sdfs)
            torch.cuda.synchronize()
    expect = (data["verts"].shape[0], data["faces"].shape[0])
    assert (verts.shape[0], faces.shape[0]) == expect, ((verts.shape[0], faces.shape[0]), expect)
    print(f"#vertices={verts.shape[0]} #triangles={faces.shape[0]}: the reference's own counts")
    out = sys.argv[1] if len(sys.argv) > 1 else "sphere_tetrahedra.ply"
    if out:
        prim3d.save_mesh(verts, faces, filename=out)
