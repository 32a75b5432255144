"""dev: wall time per whole call of the bench's call stream (no profiling hooks), configs c3 / c2-like 256^3 / c5.
usage: call_time.py [c3|c256|c5]   (P3D_CAPI_LIB / P3D_RIDE etc. from the environment)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if cfg == "c5":
    grids = torch.stack([perlin_grid(256, period=64, seed=s, device="cuda") for s in range(32)]).half()
    call = lambda: p3d.marching_cubes_batched(grids, 0.0)
    n, reps = 10, 5
else:
    N = 512 if cfg == "c3" else 256
    g = perlin_grid((N, N, N), device="cuda")
    lo, up = [0.0] * 3, [float(N)] * 3
    call = lambda: p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
    n, reps = 30, 5
for _ in range(5): out = call()
torch.cuda.synchronize()
best = []
for _ in range(reps):
    t0 = time.perf_counter()
    for _ in range(n): out = call()
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / n * 1e6)
print("%s lib=%s ride=%s: %s us per call (V=%d F=%d)" % (cfg, os.path.basename(os.environ.get("P3D_CAPI_LIB", "default")),
      os.environ.get("P3D_RIDE", "1"), " ".join("%.1f" % b for b in best), out[0].shape[0], out[1].shape[0]))
if os.environ.get("ISOLATED"):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = call(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print("   isolated calls: median %.1f us  min %.1f us" % (ts[len(ts) // 2], ts[0]))
