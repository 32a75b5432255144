#!/usr/bin/env python3
"""Counter calibration workload for tools/profile_round.sh: the scan-numbered extractor (p3d_mc_count_scan + the gather
emitter: capi.extract(with_keys=True)) on the bench grid.

Its first kernel, k_classify, reads the 512^3 fp32 field exactly once with the same dword-per-lane streaming access
shape as k_fused and writes 16 MiB of sign words, so the raw FETCH_SIZE / WRITE_SIZE it reports against those KNOWN
byte counts give the correction factor applied to k_fused's counters (MI355X_MICROARCH.md, HBM/rocprofv3 section).
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from primitive3d_amd import capi  # noqa: E402
from primitive3d_amd.fields import perlin_grid  # noqa: E402

dev = torch.device("cuda", 0)
shape = (512, 512, 512)
grid = perlin_grid(shape, period=64, seed=0, device=dev)
for _ in range(3):
    v, f = capi.extract(grid, 0.0, [0.0, 0.0, 0.0], [float(s) for s in shape], with_keys=True)[:2]
torch.cuda.synchronize()
print("calibration run:", tuple(v.shape), tuple(f.shape))
