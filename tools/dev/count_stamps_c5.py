"""dev: count_stamps.py on config c5 (32 x 256^3 fp16 through the batched entry).  Needs -DP3D_COUNT_STAMP=1 (build_dev/cstamp.so)."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
B = int(os.environ.get("B", "32"))
g = torch.stack([perlin_grid((256,) * 3, period=64, seed=s, device="cuda").half() for s in range(B)])
buf = torch.zeros((65536 * 4, 8), dtype=torch.int64, device="cuda")
L = capi.lib()
L.p3d_mc_debug_count_stamps.argtypes = [ctypes.c_void_p]
assert L.p3d_mc_debug_count_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
capi.profile_enable(2)
for _ in range(4):
    buf.zero_()
    out = p3d.marching_cubes_batched(g, 0.0)
    torch.cuda.synchronize()
    st = capi.profile_read()
print({k: round(v * 1e3, 1) for k, v in st.items()})
s = buf.cpu().numpy().astype(np.int64)
s = s[s[:, 5] > 0]
print("waves:", len(s))
tick_per_us = np.median((s[:, 5] - s[:, 0]) / np.maximum(1, (s[:, 7] - s[:, 6])) * 100.0)
print("ticks per us %.0f" % tick_per_us)
names = ["entry -> loads issued", "loads issued -> returned", "networks + wave sums", "block barrier", "offset scan + stores"]
tot = (s[:, 5] - s[:, 0]) / tick_per_us
print("wave lifetime us: median %.2f mean %.2f p10 %.2f p90 %.2f" % (np.median(tot), tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for k, nme in enumerate(names):
    d = (s[:, k + 1] - s[:, k]) / tick_per_us
    print("  %-28s median %6.2f  mean %6.2f  (%.0f %%)" % (nme, np.median(d), d.mean(), 100 * d.mean() / tot.mean()))
t0 = s[:, 6].min()
st_ = (s[:, 6] - t0) / 100.0; en = (s[:, 7] - t0) / 100.0
h, e = np.histogram(st_, bins=12)
print("wave starts histogram (us):", [(round(float(a), 1), int(b)) for a, b in zip(e[:-1], h)])
print("wave end us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print("span %.1f us; sum lifetimes/(256*16) %.1f us" % (en.max(), tot.sum() / (256 * 16)))
