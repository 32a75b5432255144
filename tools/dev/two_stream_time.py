"""dev: throughput of independent extractions issued on ONE stream back to back against TWO streams in turn (the second
call's streaming kernel beside the first call's face kernels): 512^3 Perlin, two distinct grids, pybind adapter."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", "512"))
gs = [perlin_grid((n,) * 3, period=64, seed=s, device=dev) for s in (0, 1)]
up = [float(n)] * 3
streams = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("STREAMS", "2")))]
def run(k, nstreams):
    outs = []
    for i in range(k):
        with torch.cuda.stream(streams[i % nstreams]):
            outs.append(p3d.libPrim3D.marching_cubes(gs[i % 2], 0.0, [0.0] * 3, up))
    return outs
for ns in (1, len(streams), 1, len(streams)):
    run(6, ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); o = run(40, ns); torch.cuda.synchronize()
    print("%d stream(s): %.1f us per call  V=%d" % (ns, (time.perf_counter() - t0) / 40 * 1e6, o[-1][0].shape[0]))
