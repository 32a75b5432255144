"""dev: when does marching_cubes_batched return relative to the end of its GPU work?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
B = int(os.environ.get("B", "32"))
grids = torch.stack([perlin_grid(256, period=64, seed=s, device="cuda") for s in range(B)]).half()
for _ in range(3): out = p3d.marching_cubes_batched(grids, 0.0)
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.perf_counter(); out = p3d.marching_cubes_batched(grids, 0.0); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("call returned after %.0f us; GPU finished %.0f us later" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
    time.sleep(0.01)
