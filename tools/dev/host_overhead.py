import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
g = torch.linspace(-1, 1, 8 * 8 * 64, device="cuda").reshape(8, 8, 64).contiguous()
for _ in range(20): p3d.libPrim3D.marching_cubes(g, 0.0, [0., 0., 0.], [8., 8., 64.])
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): p3d.libPrim3D.marching_cubes(g, 0.0, [0., 0., 0.], [8., 8., 64.])
torch.cuda.synchronize()
print("tiny grid whole call: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
t = time.perf_counter()
for _ in range(200):
    a = torch.empty((1000, 3), device="cuda"); b = torch.empty((1000, 3), device="cuda", dtype=torch.int32)
    c = torch.empty((100000,), device="cuda", dtype=torch.uint8); d = torch.empty((1000, 3), device="cuda")
print("4x torch.empty: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
x = torch.zeros(3, device="cuda", dtype=torch.int64)
t = time.perf_counter()
for _ in range(200): x.cpu()
print("D2H 24B + sync: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
