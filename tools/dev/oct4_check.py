"""dev: the four-octave 512^3 field behind the single-octave one (they share a shape, hence the adapter's hints): passes and
emissions per call, region totals."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
dev = torch.device("cuda", 0)
g1 = perlin_grid((512,) * 3, period=64, seed=0, device=dev)
g4 = perlin_grid((512,) * 3, period=64, seed=0, octaves=4, persistence=0.5, device=dev)
up = [512.0] * 3
def call(g):
    c0 = capi.debug_counters(); torch.cuda.synchronize(); t0 = time.perf_counter()
    v, f = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize(); t1 = time.perf_counter(); c1 = capi.debug_counters()
    return round((t1 - t0) * 1e3, 3), c1["streaming_passes"] - c0["streaming_passes"], c1["emissions_without_a_pass"] - c0["emissions_without_a_pass"], c1["count_emit_calls"] - c0["count_emit_calls"], v.shape[0]
for _ in range(5): call(g1)
print("single octave:", call(g1))
for i in range(8): print("four octaves, call", i, call(g4))
ws = torch.empty(capi.workspace_bytes(512, 512, 512), dtype=torch.uint8, device=dev)
v = torch.empty((9800000, 3), device=dev); f = torch.empty((19600000, 3), dtype=torch.int32, device=dev)
capi.extract_fused_raw(g4, 0.0, [0.0] * 3, up, ws, v, f)
print(capi.read_counts(ws, with_flags=True))
r = ws[:8192].view(torch.int64).cpu()[32:32 + 512:16]
print("regions min/max/mean", int(r.min()), int(r.max()), int(r.sum()) // 32)
