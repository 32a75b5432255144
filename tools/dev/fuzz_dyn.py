"""dev: randomized parity fuzz of the DYNAMIC plane hand-out (P3D_FUSED_DYN=2) against the oracle: shapes large enough for the
persistent blocks to get at least two planes each, fields from dense noise to a small object in a box (most blocks idle:
everything is stolen), fp32 and fp16."""
import os, sys
os.environ.setdefault("P3D_FUSED_DYN", "2")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from oracle import oracle_extract
from primitive3d_amd import capi
from bench import soup_hashes
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
gpu = torch.device("cuda", 0)
n_ok = n_dyn = 0
for it in range(int(os.environ.get("N", "30"))):
    rz = int(rng.choice([rng.integers(130, 257), rng.integers(257, 520), rng.integers(520, 1100)]))
    ry = int(rng.integers(13, 64))
    rx = int(rng.integers(2200 // max(1, (ry + 11) // 12) + 40, 1400))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        g = rng.standard_normal((rx, ry, rz)).astype(np.float32)
    elif kind == 1:
        g = rng.integers(-1, 2, size=(rx, ry, rz)).astype(np.float32)
    elif kind == 2:
        x, y, z = np.meshgrid(np.arange(rx, dtype=np.float32), np.arange(ry, dtype=np.float32), np.arange(rz, dtype=np.float32), indexing="ij")
        g = (np.sin(x * 0.07) + np.cos(y * 0.3) + np.sin(z * 0.05) + float(rng.standard_normal()) * 0.3).astype(np.float32)
    else:
        x, y, z = np.meshgrid(np.arange(rx, dtype=np.float32), np.arange(ry, dtype=np.float32), np.arange(rz, dtype=np.float32), indexing="ij")
        c = (rng.uniform(0, rx), rng.uniform(0, ry), rng.uniform(0, rz)); r = rng.uniform(5, 40)
        g = ((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2 - r * r).astype(np.float32)
    thresh = float(rng.uniform(-0.3, 0.3))
    half = bool(rng.integers(0, 2))
    t = torch.from_numpy(g).to(gpu)
    if half:
        t = t.half()
        g = t.float().cpu().numpy()
    before = capi.debug_counters()["dynamic_launches"]
    v, f = capi.extract_fused(t, thresh)
    torch.cuda.synchronize()
    n_dyn += capi.debug_counters()["dynamic_launches"] > before
    rv, rf, _ = oracle_extract(g, thresh, threads=0, want_keys=False)
    assert tuple(v.shape) == rv.shape and tuple(f.shape) == rf.shape, (it, (rx, ry, rz), kind, half, v.shape, rv.shape, f.shape, rf.shape)
    hg, kg = soup_hashes(v, f)
    ho, ko = soup_hashes(torch.from_numpy(rv).to(gpu), torch.from_numpy(rf).to(gpu))
    assert torch.equal(kg, ko) and torch.equal(hg, ho), (it, (rx, ry, rz), kind, half)
    n_ok += 1
print(f"fuzz_dyn seed {os.environ.get('SEED', '1')}: {n_ok} cases equal to the oracle, {n_dyn} of them through the dynamic hand-out")
