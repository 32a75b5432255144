"""dev: soak of the adapter's region layout -- the iso level of one field walks up and down in small steps (laid out from the last
call's totals: regions spill, rows beyond V are moved, ids translated) and jumps now and then (scratch route, spill overflow ->
re-emission): every call's counts against a table made by the two-pass pair, every 97th call's whole mesh (sorted soups)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
from bench import soup_hashes
shape = tuple(int(v) for v in os.environ.get("SHAPE", "200,192,256").split(","))
N = int(os.environ.get("N", "20000"))
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
g = perlin_grid(shape, period=32, seed=7, device="cuda", octaves=int(os.environ.get("OCT", "2")), persistence=0.5)
up = [float(s) for s in shape]
levels = [-0.06 + 0.003 * k for k in range(41)]
far = [float(v) for v in os.environ.get("FAR", "").split(",") if v]   # e.g. FAR=-0.4,-0.25,0.25,0.4: levels whose mesh is much
levels += far                                                        # smaller -- a jump back from one overflows the spill areas
table = []
for t in levels:
    v, f = capi.extract(g, t, [0.0] * 3, up)
    table.append((v.shape[0], f.shape[0], soup_hashes(v, f)))
print("levels:", [(a, b) for a, b, _ in table[:41:10]], "far:", [(a, b) for a, b, _ in table[41:]])
k = 20
c0 = capi.debug_counters()
checked = 0
t0 = time.time()
for i in range(N):
    r = rng.random()
    if r < 0.04: k = int(rng.integers(0, len(levels)))          # a jump
    elif k > 40: pass                                           # (a far level: stay until the next jump)
    elif r < 0.5: k = min(40, k + 1)
    elif r < 0.96: k = max(0, k - 1)
    v, f = p3d.libPrim3D.marching_cubes(g, levels[k], [0.0] * 3, up)
    assert (v.shape[0], f.shape[0]) == table[k][:2], (i, k, v.shape, f.shape, table[k][:2])
    if i % 97 == 0:
        assert int(f.max()) < v.shape[0]
        assert all(torch.equal(a, b) for a, b in zip(soup_hashes(v, f), table[k][2])), (i, k)
        checked += 1
torch.cuda.synchronize()
c1 = capi.debug_counters()
# (rows of 8k + 1..2 chunks take two streaming launches per pass; otherwise launches beyond one per call are re-emissions)
print("soak ok: %d calls in %.1f s, %d whole meshes compared; laid out %d, streaming launches %d" % (
    N, time.time() - t0, checked, c1["layout_passes"] - c0["layout_passes"], c1["streaming_launches"] - c0["streaming_launches"]))
