"""dev: soak -- tens of thousands of calls of every entry point; reserved device memory must stay flat and the results
stay equal to the first call's."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import primitive3d_amd as p3d
from primitive3d_amd.fields import perlin_grid
from bench import soup_hashes
dev = torch.device("cuda", 0)
g = perlin_grid((256, 256, 256), device=dev)
lo, up = [0.0] * 3, [256.0] * 3
v0, f0 = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up); torch.cuda.synchronize()
h0 = soup_hashes(v0, f0)[0]
N = int(os.environ.get("N", "30000"))
t0 = time.time(); base = None
for i in range(N):
    v, f = p3d.libPrim3D.marching_cubes(g, 0.0, lo, up)
    if i % 5000 == 0:
        torch.cuda.synchronize()
        assert v.shape == v0.shape and f.shape == f0.shape and torch.equal(soup_hashes(v, f)[0], h0), i
        r = torch.cuda.memory_reserved()
        base = base or r
        assert r <= base * 1.05, (i, r, base)
torch.cuda.synchronize()
print("marching_cubes x%d: %.1f s, reserved %.0f MiB" % (N, time.time() - t0, torch.cuda.memory_reserved() / 2**20))
gb = torch.stack([perlin_grid((128, 128, 128), seed=s, device=dev) for s in range(8)]).half()
o0 = p3d.marching_cubes_batched(gb, 0.0); torch.cuda.synchronize()
for i in range(N // 10):
    o = p3d.marching_cubes_batched(gb, 0.0)
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(o[2:], o0[2:])) and o[0].shape == o0[0].shape
print("batched x%d ok, reserved %.0f MiB" % (N // 10, torch.cuda.memory_reserved() / 2**20))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench_next
P, T, sdf = bench_next.tet_lattice(32, 0, dev, False)
tv, tf = p3d.marching_tetrahedras(P, T, sdf); torch.cuda.synchronize()
for i in range(N // 10):
    a, b = p3d.marching_tetrahedras(P, T, sdf)
torch.cuda.synchronize()
assert torch.equal(a, tv) and torch.equal(b, tf)
print("tetrahedra x%d ok, reserved %.0f MiB" % (N // 10, torch.cuda.memory_reserved() / 2**20))
rc = p3d.create_raycaster(v0 / 256.0, f0)
n = 100000
o = torch.rand(n, 3, device=dev) * 0.2 + torch.tensor([0.4, 0.4, -1.0], device=dev)
d = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0], device=dev) + 0.2 * torch.randn(n, 3, device=dev), dim=1).contiguous()
dep, nrm, ids = torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
rc.invoke(o, d, dep, nrm, ids); torch.cuda.synchronize(); d0 = dep.clone()
for i in range(N // 10):
    rc.invoke(o, d, dep, nrm, ids)
torch.cuda.synchronize()
assert torch.equal(dep, d0)
for i in range(50):   # create / destroy
    rc2 = p3d.create_raycaster(v0 / 256.0, f0); del rc2
print("ray caster x%d + 50 builds ok, reserved %.0f MiB" % (N // 10, torch.cuda.memory_reserved() / 2**20))
