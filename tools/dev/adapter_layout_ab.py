"""dev: the pybind adapter's default mode with and without the predicted region layout (P3D_MC_MODE=scratch switches it off):
back-to-back calls on the 512^3 bench grid (or GRIDS=4 distinct ones in turn), wall time per call and stage events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
ngr = int(os.environ.get("GRIDS", "1"))
octaves = int(os.environ.get("OCT", "1"))
grids = [perlin_grid((512,) * 3, period=64, seed=s, device="cuda", octaves=octaves, persistence=0.5) for s in range(ngr)]
up = [512.0] * 3
i = [0]
def call():
    g = grids[i[0] % ngr]; i[0] += 1
    return p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
hold = os.environ.get("HOLD", "1") == "1"   # 1: the caller still holds the last mesh while the next call runs (v, f = mc(...) in a loop)
if not hold:
    _call = call
    def call():
        global out
        out = None
        return _call()
for _ in range(8): out = call()
torch.cuda.synchronize()
walls = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(32): out = call()
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) / 32 * 1e6)
capi.profile_enable(2)
acc = {}
for k in range(8):
    out = call(); torch.cuda.synchronize()
    st = capi.profile_read()
    if k >= 3:
        for n, t in st.items(): acc[n] = acc.get(n, 0) + t / 5
walls.sort()
print("layout" if os.environ.get("P3D_MC_MODE") != "scratch" else "scratch", "V", out[0].shape[0], "F", out[1].shape[0],
      "call_us median %.1f min %.1f" % (walls[2], walls[0]), {n: round(t * 1e3, 1) for n, t in acc.items()})
