"""dev: one build of the C-ABI library (P3D_CAPI_LIB) on the 512^3 bench grid (or SHAPE= / SPHERE=1): per-stage hipEvent
times of the one-pass call and wall time per call of a back-to-back stream of them (raw C ABI + read_counts, the pybind
adapter's pattern).  For A/B runs of the dense-slot scheme against the scratch + copy scheme."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
n = int(os.environ.get("N", "512"))
shape = tuple(int(v) for v in os.environ["SHAPE"].split(",")) if os.environ.get("SHAPE") else (n, n, n)
if os.environ.get("SPHERE"):
    ax = [torch.arange(s, device="cuda", dtype=torch.float32) for s in shape]
    g = ((ax[0][:, None, None] - shape[0] / 4) ** 2 + (ax[1][None, :, None] - shape[1] / 4) ** 2 + (ax[2][None, None, :] - shape[2] / 4) ** 2 - (shape[0] / 8) ** 2).contiguous()
else:
    g = perlin_grid(shape, device="cuda")
if os.environ.get("HALF"):
    g = g.half()
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
capv = int(os.environ.get("CAPV", shape[0] * shape[1] * shape[2] // 16))
v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
scratch = torch.empty((capi.scratch_rows_for(capv), 3), device="cuda")
def call():
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], list(shape), ws, v, f, scratch=scratch)
    return capi.read_counts(ws, with_flags=True)
for _ in range(5): nv, nf, fl = call()
torch.cuda.synchronize()
walls = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(30): nv, nf, fl = call()
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) / 30 * 1e6)
capi.profile_enable(2)
acc = {}
for i in range(8):
    call(); torch.cuda.synchronize()
    st = capi.profile_read()
    if i >= 3:
        for k, t in st.items(): acc[k] = acc.get(k, 0) + t / 5
st = {k: round(t * 1e3, 1) for k, t in acc.items()}
st["k_sum"] = round(sum(st.values()), 1)
walls.sort()
print(os.path.basename(os.environ.get("P3D_CAPI_LIB", "default")), "V", nv, "F", nf, "flags", fl, "call_us median %.1f min %.1f" % (walls[2], walls[0]), st)
