"""dev: stage times of the one-pass call WITHOUT vertex output (no scratch, no compaction copy) next to the normal call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
g = perlin_grid(512, device="cuda")
ws = torch.empty(capi.workspace_bytes(512, 512, 512), dtype=torch.uint8, device="cuda")
capv = 512 ** 3 // 16
v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
capi.profile_enable(2)
for name, vv in (("with vertices", v), ("faces only", None), ("with vertices", v), ("faces only", None)):
    acc = {}
    for i in range(6):
        capi.extract_fused_raw(g, 0.0, [0, 0, 0], [512] * 3, ws, vv, f)
        capi.read_counts(ws); torch.cuda.synchronize()
        st = capi.profile_read()
        if i >= 2:
            for k, t in st.items(): acc[k] = acc.get(k, 0) + t / 4
    print(name, {k: round(t * 1e3, 1) for k, t in acc.items()})
