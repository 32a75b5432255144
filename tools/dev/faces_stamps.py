"""dev: where does a k_faces wave spend its life?  Needs a build with -DP3D_FACES_STAMP=1 (P3D_CAPI_LIB=build_dev/fstamp.so):
every wave leaves s_memtime stamps of its phases; prints the median / mean duration of each phase over the waves that
emitted faces, the distribution of wave lifetimes and the kernel's span."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
N = int(os.environ.get("N", "512"))
g = perlin_grid(N, device="cuda")
ws = torch.empty(capi.workspace_bytes(N, N, N), dtype=torch.uint8, device="cuda")
v = torch.empty((N ** 3 // 16, 3), device="cuda"); f = torch.empty((N ** 3 // 8, 3), dtype=torch.int32, device="cuda")
nb = (N * ((N + 63) // 64) + 255) // 256 * N      # face tiles (upper bound: tiles per plane x planes)
buf = torch.zeros((nb * 4, 16), dtype=torch.int64, device="cuda")
L = capi.lib()
L.p3d_mc_debug_face_stamps.argtypes = [ctypes.c_void_p]
assert L.p3d_mc_debug_face_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(4):
    buf.zero_()
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [N] * 3, ws, v, f)
    print(capi.read_counts(ws))
    torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64)
live = s[:, 9] > 0                        # waves that reached the end with faces to write
s = s[live]
print("waves with faces:", len(s), "of", int(live.size))
tick_per_us = np.median((s[:, 9] - s[:, 0]) / np.maximum(1, (s[:, 12] - s[:, 14])) * 100.0)   # realtime runs at 100 MHz
print("s_memtime ticks per us (median over waves): %.0f" % tick_per_us)
names = ["entry -> prologue loads issued", "loads issued -> returned", "staging compute + LDS writes", "block barrier",
         "active-cell scan + first list", "batch 0: list/windows/records/table row", "batch 0: ids computed + written",
         "batch 0: read-back + stores issued", "everything after batch 0"]
tot = (s[:, 9] - s[:, 0]) / tick_per_us
print("wave lifetime us: median %.2f  mean %.2f  p10 %.2f  p90 %.2f" % (np.median(tot), tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for k, nme in enumerate(names):
    d = (s[:, k + 1] - s[:, k]) / tick_per_us
    print("  %-44s median %6.2f us   mean %6.2f us   (%.0f %% of the mean lifetime)" % (nme, np.median(d), d.mean(), 100 * d.mean() / tot.mean()))
na = s[:, 10]; rel = s[:, 11]
print("active cells per wave: mean %.1f  median %d  max %d;  faces per wave: mean %.1f;  batches per wave: mean %.2f" %
      (na.mean(), np.median(na), na.max(), rel.mean(), np.ceil(na / 64).mean()))
span = (s[:, 12].max() - s[:, 14].min()) / 100.0
print("kernel span (first entry -> last exit, realtime): %.1f us;  sum of wave lifetimes / (256 CUs x 24 slots): %.1f us" %
      (span, tot.sum() / (256 * 24)))
# per-CU concurrency is not visible here; lifetime by batches
for nbatch in (1, 2, 3, 4):
    m = np.ceil(na / 64) == nbatch
    if m.any(): print("  waves with %d batch(es): %5d, lifetime median %.2f us, after-batch-0 median %.2f us" % (nbatch, m.sum(), np.median(tot[m]), np.median((s[m, 9] - s[m, 8]) / tick_per_us)))
