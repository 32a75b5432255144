"""dev: randomized parity fuzz of the batched (stacked) call against the oracle applied item by item."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import primitive3d_amd as p3d
from oracle import oracle_extract
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
soup = lambda vv, ff: np.sort(vv[ff.astype(np.int64)].reshape(len(ff), 9).view([("", np.float32)] * 9), axis=0)
n_ok = 0
for it in range(int(os.environ.get("N", "40"))):
    B = int(rng.integers(1, 9))
    rx, ry = int(rng.integers(2, 24)), int(rng.integers(2, 40))
    rz = int(rng.choice([rng.integers(2, 70), rng.integers(60, 140), rng.integers(190, 330), rng.integers(500, 600)]))
    dt = torch.float16 if rng.random() < 0.5 else torch.float32
    g = torch.from_numpy(rng.standard_normal((B, rx, ry, rz)).astype(np.float32) * (0.3 if rng.random() < 0.5 else 1.0)).to(dt)
    thresh = float(rng.uniform(-0.3, 0.3))
    v, f, vo, fo = p3d.marching_cubes_batched(g.cuda(), thresh)
    torch.cuda.synchronize()
    vo, fo = vo.cpu(), fo.cpu()
    for b in range(B):
        rv, rf, _ = oracle_extract(g[b].float().numpy(), thresh)
        vb, fb = v[vo[b]:vo[b + 1]].cpu().numpy(), f[fo[b]:fo[b + 1]].cpu().numpy()
        assert vb.shape == rv.shape and fb.shape == rf.shape, (it, b, (B, rx, ry, rz), vb.shape, rv.shape, fb.shape, rf.shape)
        assert np.array_equal(soup(vb, fb), soup(rv, rf)), (it, b, (B, rx, ry, rz))
    n_ok += 1
print("batched fuzz ok:", n_ok, "cases")
