"""dev: per-phase shader cycles of k_faces from a build with -DP3D_FACES_TIMING (P3D_CAPI_LIB=that build)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
g = perlin_grid(512, device="cuda")
ws = torch.zeros(capi.workspace_bytes(512, 512, 512), dtype=torch.uint8, device="cuda")
capv = 512 ** 3 // 16
v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
capi.extract_fused_raw(g, 0.0, [0, 0, 0], [512] * 3, ws, v, f)
print(capi.read_counts(ws))
torch.cuda.synchronize()
ws[:8192].zero_()
capi.extract_fused_raw(g, 0.0, [0, 0, 0], [512] * 3, ws, v, f); capi.read_counts(ws); torch.cuda.synchronize()
h = ws[:8192].view(torch.int64)[600:605].cpu().tolist()
names = ["prologue -> barrier", "unit work + cell list", "cell phase (ids)", "triangle rounds", "waves"]
tot = sum(h[:4])
for n, c in zip(names, h):
    print(f"{n:24s} {c:14d}" + (f"  {100.0 * c / tot:5.1f} %   {c / max(h[4], 1):9.0f} cycles / wave" if n != "waves" else ""))
