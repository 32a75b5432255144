import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import sphere_grid
g = torch.tensor(sphere_grid(512)).float().cuda()
shape = (512,) * 3
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
v = torch.empty((2**20, 3), device="cuda"); f = torch.empty((2**21, 3), dtype=torch.int32, device="cuda")
capi.profile_enable(2); acc = {}
for i in range(8):
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], list(shape), ws, v, f); nv, nf = capi.read_counts(ws); torch.cuda.synchronize()
    st = capi.profile_read()
    if i >= 3:
        for k, t in st.items(): acc[k] = acc.get(k, 0) + t / 5
print(os.environ.get("P3D_CAPI_LIB", "default").split("/")[-1], "sphere512 V", nv, "F", nf, {k: round(t * 1e3, 1) for k, t in acc.items()})
