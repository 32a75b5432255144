#!/bin/bash
# dev: 48-unit streaming tiles (8 chunks x 5 rows + halo row, 3 waves/SIMD) against the default 32-unit tiles
P3D_FUSED_NU48=1 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in 0 1; do echo -n "NU48=$v: "; P3D_FUSED_NU48=$v python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done
done
for v in 0 1; do echo -n "1024 NU48=$v: "; N=1024 P3D_FUSED_NU48=$v python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done
