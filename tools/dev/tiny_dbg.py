import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
g = torch.linspace(-1, 1, 8 * 8 * 64, device="cuda").reshape(8, 8, 64).contiguous()
ws = torch.empty(capi.workspace_bytes(8, 8, 64), dtype=torch.uint8, device="cuda")
v = torch.empty((4096, 3), device="cuda"); f = torch.empty((8192, 3), dtype=torch.int32, device="cuda")
capi.extract_fused_raw(g, 0.0, [0, 0, 0], [8, 8, 64], ws, v, f)
print(capi.read_counts(ws, with_flags=True))
h = ws[:8192].cpu().numpy().view("uint64")
print("cursors", [int(h[32 + 16 * r]) for r in range(32)])
