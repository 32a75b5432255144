"""dev: the bench's call stream without any profiling hooks (for a kernel trace: idle time between calls)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
g = perlin_grid((512, 512, 512), device="cuda")
lo, up = [0.0, 0.0, 0.0], [512.0, 512.0, 512.0]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # p3d_mc_profile_enable mode
read = (sys.argv[2] == "1") if len(sys.argv) > 2 else True   # read the events back after every call
for _ in range(5): out = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
torch.cuda.synchronize()
capi.profile_enable(mode)
for _ in range(12):
    out = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
    if mode and read:
        capi.profile_read()
torch.cuda.synchronize()
