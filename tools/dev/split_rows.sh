#!/bin/bash
# dev: rows of 8k + 1..4 chunks streamed by two launches (P3D_FUSED_SPLIT_ROWS=1, default) against one launch of 8-chunk tiles
RZ=513,770 N=60 SEED=61 python tools/dev/fuzz_parity.py 2>&1 | tail -1
for s in 513,511,517 512,512,600 512,512,768 384,384,640 129,1030,1100; do for v in 0 1; do echo -n "SHAPE=$s SPLIT=$v: "; SHAPE=$s P3D_FUSED_SPLIT_ROWS=$v python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done; done
P3D_FUSED_SPLIT_ROWS=1 python tools/dev/odd_shapes.py 2>&1 | tail -5
