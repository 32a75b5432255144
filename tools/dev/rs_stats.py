"""dev: life of the persistent blocks of the DYN streaming kernel.  Needs a build with -DP3D_RS_STATS=1
(P3D_CAPI_LIB=build_dev/rsstats.so): every block leaves its start / end (100 MHz ticks), the ranges it had and the planes
its first wave processed."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
N = int(os.environ.get("N", "512"))
g = perlin_grid(N, device="cuda")
ws = torch.empty(capi.workspace_bytes(N, N, N), dtype=torch.uint8, device="cuda")
v = torch.empty((N ** 3 // 16, 3), device="cuda"); f = torch.empty((N ** 3 // 8, 3), dtype=torch.int32, device="cuda")
buf = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
L = capi.lib()
L.p3d_mc_debug_rs_stats.argtypes = [ctypes.c_void_p]
assert L.p3d_mc_debug_rs_stats(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(4):
    buf.zero_()
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [N] * 3, ws, v, f)
    print(capi.read_counts(ws))
    torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64)
s = s[s[:, 1] > 0]
t0 = s[:, 0].min()
st = (s[:, 0] - t0) / 100.0; en = (s[:, 1] - t0) / 100.0
rng = s[:, 2] & 0xffffffff; pl = s[:, 2] >> 32
print("blocks", len(s), "kernel span %.1f us" % en.max())
print("start us: p0 %.1f p50 %.1f p90 %.1f max %.1f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("end   us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print("ranges per block: mean %.2f max %d;  planes per block: mean %.1f min %d max %d; total planes %d" % (rng.mean(), rng.max(), pl.mean(), pl.min(), pl.max(), pl.sum()))
print("us per plane (block life / planes): median %.2f" % np.median((en - st) / np.maximum(pl, 1)))
fd = (s[:, 4] - t0) / 100.0; stl = (s[:, 3] >> 8) / 100.0; p8 = (s[:, 5] - t0) / 100.0
print("first range done us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (fd.min(), np.percentile(fd, 10), np.median(fd), np.percentile(fd, 90), fd.max()))
print("time inside rs_switch (steal + barrier) us: p10 %.1f p50 %.1f p90 %.1f max %.1f; sum %.0f" % (np.percentile(stl, 10), np.median(stl), np.percentile(stl, 90), stl.max(), stl.sum()))
print("8th plane reached at us: p50 %.1f  -> %.2f us per plane early on;   from plane 8 to the end of the first range: %.2f us per plane" %
      (np.median(p8), np.median(p8) / 8, np.median((fd - p8) / np.maximum(1, (s[:, 2] >> 32) - 8))))
for x in range(8):
    m = (s[:, 3] & 0xff) == x
    if m.any(): print("  xcc %d: blocks %d, end median %.1f, planes mean %.1f" % (x, m.sum(), np.median(en[m]), pl[m].mean()))
late = st > 20
print("blocks starting later than 20 us:", int(late.sum()))
