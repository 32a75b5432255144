"""dev: the reference's own example sizes (66^3 bunny, 200^3 sphere) and the 256^3 bunny: wall time per call of a
back-to-back stream through the pybind adapter, and per-stage hipEvent times.  Run under rocprofv3 --kernel-trace --stats for the
kernels' own durations (GRID=bunny66|sphere200|c2)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import sphere_grid
which = os.environ.get("GRID", "bunny66")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
b66 = torch.from_numpy(np.load(os.path.join(root, "tests", "golden", "bunny66.npy")))
if which == "bunny66": g = b66.cuda().float().contiguous()
elif which == "sphere200": g = torch.tensor(sphere_grid(200)).cuda().float().contiguous()
else: g = torch.nn.functional.interpolate(b66[None, None], size=(256,) * 3, mode="trilinear", align_corners=True)[0, 0].contiguous().cuda()
up = [float(s) for s in g.shape]
call = lambda: p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
for _ in range(10): out = call()
torch.cuda.synchronize()
n = int(os.environ.get("CALLS", "200"))
t0 = time.perf_counter()
for _ in range(n): out = call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s: %.1f us per call (host loop alone %.1f us per call)  V=%d F=%d" % (which, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6, out[0].shape[0], out[1].shape[0]))
if not os.environ.get("NOSTAGES"):
    capi.profile_enable(2)
    acc = {}
    for i in range(6):
        call(); torch.cuda.synchronize()
        for k, t in capi.profile_read().items(): acc[k] = acc.get(k, 0) + t / 6
    print({k: round(t * 1e3, 1) for k, t in acc.items()}, "us (isolated calls)")
