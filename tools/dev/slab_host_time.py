"""dev: host-side wall time per piece of SlabExtractor.extract() (no profiler: perf_counter around the capi calls)."""
import os, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from primitive3d_amd import capi, slab
from primitive3d_amd.fields import perlin_grid
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    setattr(mod, name, g)
for n in ("extract_fused_raw", "read_counts", "export_plane_records", "emit", "workspace_bytes", "plane_records", "scratch_rows_for"):
    wrap(capi, n)
_empty = torch.empty
def _e(*a, **k):
    t0 = time.perf_counter(); r = _empty(*a, **k); dt = time.perf_counter() - t0
    key = "empty %s%s" % ("pinned " if k.get("pin_memory") else "", "x".join(str(int(v)) for v in (a[0] if a and isinstance(a[0], (tuple, list)) else a)) if r.numel() < 1000 else "%d MB" % (r.numel() * r.element_size() >> 20))
    acc[key] += dt; cnt[key] += 1; return r
torch.empty = _e
ext = slab.SlabExtractor.extract
def timed_extract(self, *a, **k):
    t0 = time.perf_counter(); r = ext(self, *a, **k); acc["extract() total"] += time.perf_counter() - t0; cnt["extract() total"] += 1; return r
slab.SlabExtractor.extract = timed_extract
r = bench.rank_slab_workload(capi, perlin_grid, torch.device("cuda", 0), steps=300, warmup=10)
print("ms per step", r["ms_per_step"])
for k in sorted(acc, key=lambda k: -acc[k]):
    print("%-24s %6d calls  %8.1f us per call   %8.1f us per step" % (k, cnt[k], acc[k] / cnt[k] * 1e6, acc[k] / cnt["extract() total"] * 1e6))
