"""dev: randomized parity fuzz of the one-pass call, the scan-numbered pair and the two-pass pair (p3d_mc_count +
p3d_mc_emit) against the oracle (small random shapes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from oracle import oracle_extract
from tests.test_gpu_parity import _assert_same_mesh, _hip_extract, _hip_extract_fused, _hip_extract_pair
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
gpu = torch.device("cuda", 0)
n_ok = 0
for it in range(int(os.environ.get("N", "60"))):
    rx, ry = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    rz = int(rng.choice([rng.integers(2, 70), rng.integers(60, 140), rng.integers(129, 257), rng.integers(250, 700),
                         rng.integers(2040, 2200)]))
    if os.environ.get("RZ"):   # e.g. RZ=129,257: only rows of that length range
        lo, hi = (int(v) for v in os.environ["RZ"].split(","))
        rz = int(rng.integers(lo, hi))
    kind = rng.integers(0, 3)
    if kind == 0:
        g = rng.standard_normal((rx, ry, rz)).astype(np.float32)
    elif kind == 1:
        g = rng.integers(-2, 3, size=(rx, ry, rz)).astype(np.float32)
    else:
        x, y, z = np.meshgrid(np.arange(rx), np.arange(ry), np.arange(rz), indexing="ij")
        g = (np.sin(x * 0.7) + np.cos(y * 0.5) + np.sin(z * 0.11) + rng.standard_normal() * 0.3).astype(np.float32)
    thresh = float(rng.uniform(-0.5, 0.5))
    dt = torch.float32
    if os.environ.get("DTYPE") == "f16":   # fp16 grid: the oracle sees the up-cast values (marching_cubes.py:87)
        g = g.astype(np.float16)
        ref = oracle_extract(g.astype(np.float32), thresh)
        dt = torch.float16
    else:
        ref = oracle_extract(g, thresh)
    _assert_same_mesh(_hip_extract_fused(gpu, g, thresh, None, None, dtype=dt), ref)
    _assert_same_mesh(_hip_extract(gpu, g, thresh, None, None, dtype=dt), ref)
    _assert_same_mesh(_hip_extract_pair(gpu, g, thresh, None, None, dtype=dt), ref)   # (count-only pass + second streaming pass)
    n_ok += 1
print("fuzz ok:", n_ok, "cases")
