"""dev: a grid above 4 GiB (plane offsets beyond 32 bits): counts against the independent torch count, mesh properties."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
from tests.test_gpu_configs import mesh_properties, torch_counts
shape = tuple(int(v) for v in os.environ.get("SHAPE", "1280,1024,1024").split(","))
g = perlin_grid(shape, period=64, seed=1, device="cuda")
lo, up = [0.0, 0.0, 0.0], [float(s) for s in shape]
for _ in range(2): v, f = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): v, f = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
torch.cuda.synchronize(); t1 = time.perf_counter()
want = torch_counts(g, 0.0)
assert (v.shape[0], f.shape[0]) == want, (v.shape, f.shape, want)
mesh_properties(v, f)
n = shape[0] * shape[1] * shape[2]
print(shape, "%.2f GiB ok  %.3f ms  %.0f Mvoxels/s  V=%d F=%d" % (n * 4 / 2**30, (t1 - t0) / 5 * 1e3, n * 5 / (t1 - t0) / 1e6, v.shape[0], f.shape[0]))
