"""dev: config 5 (32 x 256^3 fp16) through marching_cubes_batched."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
B = int(os.environ.get("B", "32"))
grids = torch.stack([perlin_grid(256, period=64, seed=s, device="cuda") for s in range(B)])
if not os.environ.get("F32"): grids = grids.half()
for _ in range(2): out = p3d.marching_cubes_batched(grids, 0.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out = p3d.marching_cubes_batched(grids, 0.0)
torch.cuda.synchronize(); t1 = time.perf_counter()
dt = (t1 - t0) / 5
print("batch of %d x 256^3 fp16: %.2f ms  (%.1f us/item, %.0f Mvoxels/s)  V=%d F=%d" % (B, dt * 1e3, dt / B * 1e6, B * 256**3 / dt / 1e6, out[0].shape[0], out[1].shape[0]))
from primitive3d_amd import capi
capi.profile_enable(2)
out = p3d.marching_cubes_batched(grids, 0.0); torch.cuda.synchronize()
print({k: round(t * 1e3, 1) for k, t in capi.profile_read().items()}, "us")
