"""dev: odd (non power-of-two) shapes at full size: counts against an independent torch count, mesh properties, time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
from tests.test_gpu_configs import mesh_properties, torch_counts
for shape in [(500, 500, 500), (513, 511, 517), (300, 700, 450), (129, 1030, 1100), (1025, 65, 2050)]:
    g = perlin_grid(shape, period=48, seed=3, device="cuda")
    lo, up = [0.0, 0.0, 0.0], [float(s) for s in shape]
    for _ in range(3): v, f = p3d.libPrim3D.marching_cubes(g, 0.01, lo, up)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): v, f = p3d.libPrim3D.marching_cubes(g, 0.01, lo, up)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    assert (v.shape[0], f.shape[0]) == torch_counts(g, 0.01), (shape, v.shape, f.shape, torch_counts(g, 0.01))
    mesh_properties(v, f)
    n = shape[0] * shape[1] * shape[2]
    print(shape, "ok  %.1f us  %.0f Mvoxels/s  V=%d F=%d" % ((t1 - t0) / 10 * 1e6, n * 10 / (t1 - t0) / 1e6, v.shape[0], f.shape[0]))
    del g, v, f
