"""dev: does a short idle period (a 1-wave spin kernel) before each call bring the kernels back to their isolated
durations?  (kernel trace: compare k_fused / k_faces averages with and without the spin).  argv[1] = spin cycles (0 = none)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
g = perlin_grid((512, 512, 512), device="cuda")
lo, up = [0.0, 0.0, 0.0], [512.0, 512.0, 512.0]
spin = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(5): out = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
torch.cuda.synchronize()
for _ in range(30):
    if spin:
        torch.cuda._sleep(spin)
    out = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
torch.cuda.synchronize()
