"""dev: lives of the streaming kernel's blocks (fixed slabs or DYN; needs -DP3D_RS_STATS=1, P3D_CAPI_LIB=build_dev/rsstats.so):
how many blocks are alive over time -- where the launch's end loses its occupancy."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
N = int(os.environ.get("N", "512"))
shape = tuple(int(v) for v in os.environ["SHAPE"].split(",")) if os.environ.get("SHAPE") else (N, N, N)
nvox = shape[0] * shape[1] * shape[2]
g = perlin_grid(shape, device="cuda")
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
v = torch.empty((nvox // 16, 3), device="cuda"); f = torch.empty((nvox // 8, 3), dtype=torch.int32, device="cuda")
buf = torch.zeros((16384, 8), dtype=torch.int64, device="cuda")
L = capi.lib()
L.p3d_mc_debug_rs_stats.argtypes = [ctypes.c_void_p]
assert L.p3d_mc_debug_rs_stats(ctypes.c_void_p(buf.data_ptr())) == 0
capi.profile_enable(2)
for _ in range(4):
    buf.zero_()
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], list(shape), ws, v, f)
    capi.read_counts(ws)
    torch.cuda.synchronize()
    st = capi.profile_read()
s = buf.cpu().numpy().astype(np.int64)
s = s[s[:, 1] > 0]
t0 = s[:, 0].min()
a = (s[:, 0] - t0) / 100.0; b = (s[:, 1] - t0) / 100.0
pl = s[:, 2] >> 32
print("k_fused by events %.1f us; blocks %d; span %.1f us; planes per block: mean %.1f" % (st["k_fused"] * 1e3, len(s), b.max(), pl.mean()))
print("block life us: median %.1f p10 %.1f p90 %.1f;  us per plane (life / planes incl. halo): median %.2f" %
      (np.median(b - a), np.percentile(b - a, 10), np.percentile(b - a, 90), np.median((b - a) / np.maximum(1, pl + 1))))
edges = np.arange(0, b.max() + 5, 5.0)
alive = [(int(((a < t + 5) & (b > t)).sum()), float(np.clip(np.minimum(b, t + 5) - np.maximum(a, t), 0, None).sum() / 5.0)) for t in edges]
print("time window us : blocks alive (time-weighted)")
for t, (n, w) in zip(edges, alive):
    print("  %5.0f - %5.0f : %6.1f %s" % (t, t + 5, w, "#" * int(w / 16)))
print("slot-time used: %.0f block-us = %.1f us x 1024 slots; idle slot-time inside the span: %.1f us x 1024" %
      ((b - a).sum(), (b - a).sum() / 1024, b.max() - (b - a).sum() / 1024))

# long blocks only: how long a plane takes as a function of WHEN the block ran
big = pl >= max(4, int(np.percentile(pl, 90)) - 1)
print("blocks of >= %d planes: us per plane load (life / (planes + 1)) by start time" % pl[big].min())
w = float(os.environ.get("BUCKET", "10"))
for t in np.arange(0, a[big].max() + w, w):
    m = big & (a >= t) & (a < t + w)
    if m.sum():
        r = (b[m] - a[m]) / (pl[m] + 1)
        print("  start %5.0f - %5.0f : %5d blocks  median %.2f  p10 %.2f  p90 %.2f" % (t, t + w, m.sum(), np.median(r), np.percentile(r, 10), np.percentile(r, 90)))
