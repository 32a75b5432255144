"""The reference's acceptance script (examples/sphere.py) on the MI355X build: 200^3 int64 sphere,
GPU extraction through `prim3d.marching_cubes`, PLY export, counts asserted against the reference's
known answer (V=11766, F=23528; the reference compares with PyMCubes, which is used here only if
installed)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))  # run from a source checkout
import prim3d  # noqa: E402

X, Y, Z = np.mgrid[:200, :200, :200]
DENSITY_GRID = (X - 50) ** 2 + (Y - 50) ** 2 + (Z - 50) ** 2 - 25 ** 2

if __name__ == "__main__":
    density_grid_cu = torch.tensor(DENSITY_GRID).cuda()
    with prim3d.Timer("cuda marching cubes: {:.6f}s"):
        vertices_cu, faces_cu = prim3d.marching_cubes(density_grid_cu, 0, verbose=True)
    with prim3d.Timer("prim3d save mesh: {:.6f}s\n"):
        prim3d.save_mesh(vertices_cu, faces_cu, filename="sphere.ply")
    assert vertices_cu.shape[0] == 11766 and faces_cu.shape[0] == 23528
    try:
        import mcubes
        with prim3d.Timer("cpu marching cubes: {:.6f}s"):
            vertices_c, faces_c = mcubes.marching_cubes(DENSITY_GRID, 0)
        assert vertices_cu.shape[0] == vertices_c.shape[0] and faces_cu.shape[0] == faces_c.shape[0]
    except ImportError:
        print("mcubes not installed: compared with the recorded reference counts only")
