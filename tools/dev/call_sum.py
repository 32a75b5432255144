"""dev: like fused_time.py but prints the SUM of the three kernels too (stage split experiments)."""
import os, sys, runpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
g = perlin_grid((512,) * 3, device="cuda")
ws = torch.empty(capi.workspace_bytes(512, 512, 512), dtype=torch.uint8, device="cuda")
capv = 512 ** 3 // 16
v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
capi.profile_enable(2)
acc = {}
for i in range(10):
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [512] * 3, ws, v, f)
    capi.read_counts(ws); torch.cuda.synchronize()
    st = capi.profile_read()
    if i >= 3:
        for k, t in st.items(): acc[k] = acc.get(k, 0) + t / 7
acc["k_sum"] = sum(acc.values())
print({k: round(t * 1e3, 1) for k, t in acc.items()})
