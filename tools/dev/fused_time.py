"""dev: time the one-pass kernel alone (through the C ABI) on the 512^3 Perlin grid."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
n = int(os.environ.get("N", "512"))
shape = tuple(int(v) for v in os.environ["SHAPE"].split(",")) if os.environ.get("SHAPE") else (n, n, n)
g = perlin_grid(shape, device="cuda")
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
capv = shape[0] * shape[1] * shape[2] // 16
v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
capi.profile_enable(2)
acc = {}
flush = torch.empty(int(os.environ.get("FLUSH_MB", "0")) << 20, dtype=torch.uint8, device="cuda")   # FLUSH_MB: written between calls (what the
for i in range(8):                                                                                   # memory-side cache keeps of the grid from call to call)
    if flush.numel(): sink = flush.view(torch.int32).sum() if os.environ.get("FLUSH_READ", "1") != "0" else flush.fill_(i)   # (a READ: clean lines, no write-back traffic under the next kernel)
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], list(shape), ws, v, f)
    nv, nf = capi.read_counts(ws)
    torch.cuda.synchronize()
    st = capi.profile_read()
    if i >= 3:
        for k, t in st.items(): acc[k] = acc.get(k, 0) + t / 5
print(os.environ.get("P3D_CAPI_LIB", "default"), "V", nv, "F", nf, {k: round(t * 1e3, 1) for k, t in acc.items()}, "us")
