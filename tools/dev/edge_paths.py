"""dev: the adapter's rarely taken paths against the oracle -- a too-small output guess with faces and without (flat grids),
a field denser than the scratch, and (P3D_TEST_ID_LIMIT=...) the renumbering fallback; any P3D_MC_MODE."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
from bench import soup_hashes
from oracle import oracle_extract
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
cases = [("flat noise 1x300x300", torch.from_numpy(rng.standard_normal((1, 300, 300)).astype(np.float32)).to(dev)),
         ("line 1x1x5000", torch.from_numpy(rng.standard_normal((1, 1, 5000)).astype(np.float32)).to(dev)),
         ("white noise 40x50x70", torch.from_numpy(rng.standard_normal((40, 50, 70)).astype(np.float32)).to(dev)),
         ("perlin p8 96x96x130", perlin_grid((96, 96, 130), period=8, seed=2, device=dev)),
         ("perlin p16 150x140x200", perlin_grid((150, 140, 200), period=16, seed=3, device=dev))]
for name, g in cases:
    up = [float(s) for s in g.shape]
    ov, of = oracle_extract(g.cpu().numpy(), 0.0, [0.0] * 3, up)[:2]
    for k in range(2):
        c0 = capi.debug_counters()
        v, f = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
        torch.cuda.synchronize()
        c1 = capi.debug_counters()
        assert v.shape[0] == ov.shape[0] and f.shape[0] == of.shape[0], (name, k, v.shape, ov.shape, f.shape, of.shape)
        vs = torch.sort(v.contiguous().view(torch.int32).long().mul(torch.tensor([1, 1 << 21, 1 << 42], device=dev)).sum(1)).values
        os_ = torch.sort(torch.from_numpy(ov).to(dev).view(torch.int32).long().mul(torch.tensor([1, 1 << 21, 1 << 42], device=dev)).sum(1)).values
        assert torch.equal(vs, os_), (name, k, "vertex sets differ")
        if f.shape[0]:
            a = soup_hashes(v, f); b = soup_hashes(torch.from_numpy(ov).to(dev), torch.from_numpy(of.astype(np.int32)).to(dev))
            assert torch.equal(a[0], b[0]), (name, k)
        print("%-26s call %d ok  V %8d F %8d  passes %d  emissions w/o pass %d  count+emit %d" % (
            name, k, v.shape[0], f.shape[0], c1["streaming_passes"] - c0["streaming_passes"],
            c1["emissions_without_a_pass"] - c0["emissions_without_a_pass"], c1["count_emit_calls"] - c0["count_emit_calls"]))
