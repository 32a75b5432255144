import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
import primitive3d_amd as p3d
g = torch.linspace(-1, 1, 8 * 8 * 64, device="cuda").reshape(8, 8, 64).contiguous()
for _ in range(5): p3d.libPrim3D.marching_cubes(g, 0.0, [0., 0., 0.], [8., 8., 64.])
capi.profile_enable(2)
p3d.libPrim3D.marching_cubes(g, 0.0, [0., 0., 0.], [8., 8., 64.])
torch.cuda.synchronize()
print({k: round(v * 1e3, 1) for k, v in capi.profile_read().items()})
