"""dev: unusual fields / shapes through the adapter: counts vs an independent torch count, time per call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid, sphere_grid
from tests.test_gpu_configs import torch_counts
def run(name, g, thresh):
    shape = tuple(g.shape); lo, up = [0.0, 0.0, 0.0], [float(s) for s in shape]
    times = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v, f = p3d.libPrim3D.marching_cubes(g, thresh, lo, up)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e6)
    assert (v.shape[0], f.shape[0]) == torch_counts(g, thresh), (name, v.shape, f.shape, torch_counts(g, thresh))
    print("%-28s %-18s V=%-9d F=%-9d first calls %s us, steady %.0f us" % (name, shape, v.shape[0], f.shape[0], [int(t) for t in times[:3]], min(times[3:])))
dev = "cuda"
run("small sphere in 512^3", torch.tensor(sphere_grid(512)).float().to(dev) + 3000.0, 0.0)  # radius shrunk by the offset
run("sphere 512^3", torch.tensor(sphere_grid(512)).float().to(dev), 0.0)
run("white noise 256^3", torch.randn(256, 256, 256, device=dev), 0.0)
run("thin 2x2048x2048", perlin_grid((2, 2048, 2048), period=64, seed=1, device=dev), 0.0)
run("thin 2048x2048x2", perlin_grid((2048, 2048, 2), period=64, seed=1, device=dev), 0.0)
run("long rows 8x8x200000", perlin_grid((8, 8, 200000), period=64, seed=1, device=dev), 0.0)
run("all outside 512^3", torch.zeros(512, 512, 512, device=dev), 0.5)
