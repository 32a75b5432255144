"""dev: how far the 32 region totals move between two fields (sum |delta| / V): the adapter lays the next call out from the
last one's totals when this is below 1/64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
def regions(g, t=0.0):
    shape = g.shape
    ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
    capv = shape[0] * shape[1] * shape[2] // 8
    v = torch.empty((capv, 3), device="cuda"); f = torch.empty((2 * capv, 3), dtype=torch.int32, device="cuda")
    capi.extract_fused_raw(g, t, [0, 0, 0], [float(s) for s in shape], ws, v, f)
    nv, nf, fl, reg = capi.read_counts_ex(ws)
    return nv, list(reg)
def drift(a, b):
    return sum(abs(x - y) for x, y in zip(a[1], b[1])) / b[0]
for oct_ in (1, 4):
    gs = [perlin_grid((512,) * 3, period=64, seed=s, device="cuda", octaves=oct_, persistence=0.5) for s in range(4)]
    r = [regions(g) for g in gs]
    print("octaves", oct_, "seed to seed:", ["%.4f" % drift(r[i], r[(i + 1) % 4]) for i in range(4)])
    r = [regions(gs[0], 0.003 * k) for k in range(4)]
    print("octaves", oct_, "iso level +0.003 per call:", ["%.4f" % drift(r[i], r[i + 1]) for i in range(3)])
    r = [regions(gs[0], 0.01 * k) for k in range(4)]
    print("octaves", oct_, "iso level +0.01 per call:", ["%.4f" % drift(r[i], r[i + 1]) for i in range(3)])
