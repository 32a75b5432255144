#!/bin/bash
# dev: 16-unit streaming tiles for a single small grid (rz <= 256) against the half-empty 8-chunk tile
P3D_FUSED_SMALL16=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "not 1024 and not c5" 2>&1 | tail -2
for i in 1 2 3; do for v in 0 1; do for n in 256 128; do echo -n "N=$n SMALL16=$v: "; N=$n P3D_FUSED_SMALL16=$v python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done; done; done
for v in 0 1; do echo -n "SHAPE=512,512,256 SMALL16=$v: "; SHAPE=512,512,256 P3D_FUSED_SMALL16=$v python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done
for v in 0 1; do echo "c2 SMALL16=$v: "; P3D_FUSED_SMALL16=$v python bench.py --config c2 --steps 50 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330; done
