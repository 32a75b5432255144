"""dev: plain one-pass call on the per-rank shapes of the weak-scaling bench (no slab machinery)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
for shape in [(512, 512, 512), (256, 1024, 512), (128, 1024, 1024), (1024, 512, 512)]:
    g = perlin_grid(shape, period=64, seed=0, device="cuda")
    lo, up = [0.0, 0.0, 0.0], [float(s) for s in shape]
    for _ in range(5): out = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    n = shape[0] * shape[1] * shape[2]
    print(shape, "%.1f us/call  %.0f Mvoxels/s  V=%d F=%d" % ((t1 - t0) / 20 * 1e6, n * 20 / (t1 - t0) / 1e6, out[0].shape[0], out[1].shape[0]))
    del g, out
