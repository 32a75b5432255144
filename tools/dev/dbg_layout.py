import sys; sys.path.insert(0, "/root/repo")
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
shape = (33, 30, 1100)
g = perlin_grid(shape, period=14, seed=sum(shape)).cuda()
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
capv = max(4096, g.numel() // 4)
v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
capi.extract_fused_raw(g, 0.03, [0.5, -1.0, 2.0], [3.0, 4.0, 9.0], ws, v, f)
print(capi.read_counts_ex(ws), capv, capi.scratch_rows_for(capv) // 32)
hdr = ws[:8192].view(torch.int64).cpu()
print("hdr V F flags recform", hdr[:4].tolist(), "cursors", hdr[32:32+512:16].tolist())
