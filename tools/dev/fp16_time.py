import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
g = perlin_grid(512, device="cuda").half()
ws = torch.empty(capi.workspace_bytes(512, 512, 512), dtype=torch.uint8, device="cuda")
v = torch.empty((8 * 2**20, 3), device="cuda"); f = torch.empty((16 * 2**20, 3), dtype=torch.int32, device="cuda")
capi.profile_enable(2)
for i in range(5):
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [512] * 3, ws, v, f)
    print(capi.read_counts(ws), {k: round(t * 1e3, 1) for k, t in capi.profile_read().items()})
