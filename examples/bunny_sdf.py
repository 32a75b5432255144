"""The reference's examples/bunny_sdf.py on the MI355X build (input: the reference's bunny.npy, kept as
tests/golden/bunny66.npy; known answer V=13282, F=26560)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))  # run from a source checkout
import prim3d  # noqa: E402

DENSITY_GRID = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bunny66.npy"))
print(f"DENSITY_GRID shape: ({DENSITY_GRID.shape[0]}, {DENSITY_GRID.shape[1]}, {DENSITY_GRID.shape[2]})")

if __name__ == "__main__":
    density_grid_cu = torch.tensor(DENSITY_GRID).cuda()
    with prim3d.Timer("cuda marching cubes: {:.6f}s"):
        vertices_cu, faces_cu = prim3d.marching_cubes(density_grid_cu, 0, verbose=True)
    with prim3d.Timer("prim3d save mesh: {:.6f}s\n"):
        prim3d.save_mesh(vertices_cu, faces_cu, filename="bunny.ply")
    assert vertices_cu.shape[0] == 13282 and faces_cu.shape[0] == 26560
