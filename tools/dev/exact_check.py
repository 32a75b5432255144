"""dev: the adapter on a few fields incl. degenerate shapes, against the oracle; P3D_MC_MODE=exact also checks the storage sizes.  Ends with the 512^3 call time."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid, sphere_grid
from bench import soup_hashes
from oracle import oracle_extract
dev = torch.device("cuda", 0)
for name, g in (("noise", perlin_grid((70, 64, 130), period=16, seed=5, device=dev)), ("sphere", torch.tensor(sphere_grid(96)).float().cuda()),
                ("empty", torch.ones((40, 40, 70), device=dev)), ("flat", perlin_grid((1, 50, 130), period=8, seed=3, device=dev)),
                ("tiny", perlin_grid((2, 2, 2), period=2, seed=1, device=dev)), ("line", perlin_grid((1, 1, 300), period=8, seed=4, device=dev)), ("noise2", perlin_grid((200, 190, 260), period=24, seed=2, device=dev))):
    up = [float(s) for s in g.shape]
    p0 = capi.debug_counters()["streaming_passes"]
    for _ in range(3):
        v, f = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()
    passes = capi.debug_counters()["streaming_passes"] - p0
    ov, of = oracle_extract(g.cpu().numpy(), 0.0, [0.0] * 3, up)[:2]
    assert v.shape[0] == ov.shape[0] and f.shape[0] == of.shape[0], (name, v.shape, ov.shape)
    if os.environ.get("P3D_MC_MODE") == "exact":
        assert v.untyped_storage().nbytes() == max(0, v.shape[0]) * 12 and f.untyped_storage().nbytes() == f.shape[0] * 12, name
    if f.shape[0]:
        a = soup_hashes(v, f); b = soup_hashes(torch.from_numpy(ov).cuda(), torch.from_numpy(of.astype(np.int32)).cuda())
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), name
    print(name, "ok V", v.shape[0], "F", f.shape[0], "passes for 3 calls:", passes)
g = perlin_grid((512, 512, 512), period=64, seed=0, device=dev)
up = [512.0] * 3
for _ in range(3): out = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
torch.cuda.synchronize(); print("512^3 exact mode: %.1f us per call" % ((time.perf_counter() - t0) / 10 * 1e6), out[0].shape, out[1].shape)
