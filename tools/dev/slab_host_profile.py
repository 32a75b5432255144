"""dev: where the HOST spends its time in SlabExtractor.extract() (cProfile over the timed steps only; tottime per step)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from primitive3d_amd import capi, slab
from primitive3d_amd.fields import perlin_grid
pr = cProfile.Profile()
ext = slab.SlabExtractor.extract
n = [0]
def prof_extract(self, *a, **k):
    n[0] += 1
    if n[0] <= 20:
        return ext(self, *a, **k)
    pr.enable()
    try:
        return ext(self, *a, **k)
    finally:
        pr.disable()
slab.SlabExtractor.extract = prof_extract
r = bench.rank_slab_workload(capi, perlin_grid, torch.device("cuda", 0), steps=300, warmup=10)
steps = n[0] - 20
print("ms per step (profiled)", r["ms_per_step"], "steps profiled", steps)
st = pstats.Stats(pr)
rows = []
for (fn, ln, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / steps * 1e6, ct / steps * 1e6, nc / steps, "%s:%d(%s)" % (os.path.basename(fn), ln, name)))
rows.sort(reverse=True)
print("%9s %9s %7s  function" % ("tot us", "cum us", "calls"))
for tt, ct, nc, nm in rows[:34]:
    print("%9.1f %9.1f %7.1f  %s" % (tt, ct, nc, nm))
