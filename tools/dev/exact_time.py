"""dev: wall time per call of the adapter on the 512^3 bench grid, back-to-back calls (P3D_MC_MODE=exact for the exact mode)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
g = perlin_grid((512, 512, 512), period=64, seed=0, device=torch.device("cuda", 0))
up = [512.0] * 3
for _ in range(5): out = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
print("mode=%s ring=%s: %s us per call (5 x 20 calls)" % (os.environ.get("P3D_MC_MODE", "default"), os.environ.get("P3D_PARTS_RING", "1"), " ".join("%.1f" % t for t in ts)))
