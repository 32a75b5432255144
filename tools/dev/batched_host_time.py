"""dev: host-side wall time per piece of marching_cubes_batched on config c5, and when the host is ready for the next call."""
import os, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    setattr(mod, name, g)
for n in ("extract_fused_batched_raw", "read_counts", "workspace_bytes_batched"):
    wrap(capi, n)
_empty = torch.empty
def _e(*a, **k):
    t0 = time.perf_counter(); r = _empty(*a, **k); acc["torch.empty"] += time.perf_counter() - t0; cnt["torch.empty"] += 1; return r
torch.empty = _e
g = torch.stack([perlin_grid((256,) * 3, period=64, seed=s, device="cuda").half() for s in range(32)])
for _ in range(3): out = p3d.marching_cubes_batched(g, 0.0)
torch.cuda.synchronize(); acc.clear(); cnt.clear()
N = 20; t0 = time.perf_counter()
for _ in range(N):
    t1 = time.perf_counter(); out = p3d.marching_cubes_batched(g, 0.0); acc["call total"] += time.perf_counter() - t1
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print("ms per step %.4f" % (dt * 1e3))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("%-28s %5.1f calls/step  %8.1f us per step" % (k, cnt[k] / N if cnt[k] else 1, acc[k] / N * 1e6))
