import sys, os, ctypes; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
shape = (512, 512, 512)
g = perlin_grid(shape, period=64, seed=0, device="cuda")
up = [512.0] * 3
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
capv0 = 512 ** 3 // 16
v0 = torch.empty((capv0, 3), device="cuda"); f = torch.empty((2 * capv0, 3), dtype=torch.int32, device="cuda")
capi.extract_fused_raw(g, 0.0, [0, 0, 0], up, ws, v0, f)
nv, nf, fl, reg = capi.read_counts_ex(ws)
print("scratch", nv, nf, fl)
first, rows = capi.region_layout(reg)
v = torch.empty((rows, 3), device="cuda")
slab = capi.Slab(); slab.region_first_rows = ctypes.cast(first, ctypes.c_void_p)
ws[900 * 8:902 * 8].zero_()
capi.extract_fused_raw(g, 0.0, [0, 0, 0], up, ws, v, f, slab=slab)
nv2, nf2, fl2, reg2 = capi.read_counts_ex(ws)
torch.cuda.synchronize()
print("layout", nv2, nf2, fl2, reg2 == reg, "rows", rows)
lay = capi.debug_layout(*shape)
rec = ws[lay["off_records"]:lay["off_records"] + lay["num_units"] * 8].view(torch.int32).view(-1, 2)[:, 0].long() & 0xffffffff
hdr = ws[:8192].view(torch.int64).cpu()
V, end = int(hdr[0]), int(hdr[600 + 33])
hit = (rec + 191 >= V) & (rec < end)
print("V", V, "layout end", end, "units", rec.numel(), "records in the window", int(hit.sum()), "of which >= V", int(((rec >= V) & (rec < end)).sum()))
print("spill cursor / occ[32]", int(hdr[840 + 32]), "holes", int(hdr[760 + 33]), "recform", int(hdr[3]))
vals = rec[hit][:20].tolist(); print(vals)
print("blocks on the per-id path", int(hdr[900]), "waves on the tail path", int(hdr[901]))
