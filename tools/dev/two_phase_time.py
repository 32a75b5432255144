"""dev: the literal two-phase binding (p3d_mc_count -> read_counts -> exact tensors -> p3d_mc_emit) on the 512^3 bench grid:
wall time per call of a back-to-back stream, and the stage events of both calls."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
n = int(os.environ.get("N", "512"))
g = perlin_grid((n, n, n), device="cuda")
for _ in range(3): v, f = capi.extract(g, 0.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): v, f = capi.extract(g, 0.0)
torch.cuda.synchronize()
print("two-phase: %.1f us per call  V=%d F=%d" % ((time.perf_counter() - t0) / 10 * 1e6, v.shape[0], f.shape[0]))
capi.profile_enable(2)
v, f = capi.extract(g, 0.0); torch.cuda.synchronize()
print({k: round(t * 1e3, 1) for k, t in capi.profile_read().items()})
