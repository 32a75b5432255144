"""dev: where does a k_face_count_walk wave spend its life?  Needs a build with -DP3D_COUNT_STAMP=1
(P3D_CAPI_LIB=build_dev/cstamp.so): every wave leaves s_memtime stamps of its phases."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
N = int(os.environ.get("N", "512"))
g = perlin_grid(N, device="cuda")
ws = torch.empty(capi.workspace_bytes(N, N, N), dtype=torch.uint8, device="cuda")
v = torch.empty((N ** 3 // 16, 3), device="cuda"); f = torch.empty((N ** 3 // 8, 3), dtype=torch.int32, device="cuda")
buf = torch.zeros((8192 * 4, 8), dtype=torch.int64, device="cuda")
L = capi.lib()
L.p3d_mc_debug_count_stamps.argtypes = [ctypes.c_void_p]
assert L.p3d_mc_debug_count_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
capi.profile_enable(2)
for _ in range(4):
    buf.zero_()
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [N] * 3, ws, v, f)
    capi.read_counts(ws)
    torch.cuda.synchronize()
    st = capi.profile_read()
print("k_face_count_walk by events: %.1f us" % (st["k_face_count_walk"] * 1e3))
s = buf.cpu().numpy().astype(np.int64)
s = s[s[:, 5] > 0]
print("waves:", len(s))
tick_per_us = np.median((s[:, 5] - s[:, 0]) / np.maximum(1, (s[:, 7] - s[:, 6])) * 100.0)   # realtime runs at 100 MHz
print("s_memtime ticks per us (median over waves): %.0f" % tick_per_us)
names = ["entry -> loads issued", "loads issued -> returned", "first-bit words, activity test, networks, wave sums", "block barrier",
         "offset scan by wave 0, stores (+ second barrier)"]
tot = (s[:, 5] - s[:, 0]) / tick_per_us
print("wave lifetime us: median %.2f  mean %.2f  p10 %.2f  p90 %.2f" % (np.median(tot), tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for k, nme in enumerate(names):
    d = (s[:, k + 1] - s[:, k]) / tick_per_us
    print("  %-52s median %6.2f us   mean %6.2f us   (%.0f %% of the mean lifetime)" % (nme, np.median(d), d.mean(), 100 * d.mean() / tot.mean()))
t0 = s[:, 6].min()
st_ = (s[:, 6] - t0) / 100.0; en = (s[:, 7] - t0) / 100.0
print("wave start us: p0 %.1f p50 %.1f p90 %.1f max %.1f;   wave end us: p10 %.1f p50 %.1f p90 %.1f max %.1f" %
      (st_.min(), np.median(st_), np.percentile(st_, 90), st_.max(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print("kernel span (first entry -> last exit, realtime): %.1f us;  sum of wave lifetimes / (256 CUs x 16 slots): %.1f us" % (en.max(), tot.sum() / (256 * 16)))
