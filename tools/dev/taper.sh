#!/bin/bash
# dev: taper variants of the streaming kernel, interleaved on one box
for i in 1 2 3; do
  for cfg in "4 2" "8 2" "8 4" "4 4" "16 4" "1000 8"; do
    set -- $cfg
    echo -n "TAIL_DIV=$1 XT_TAIL=$2: "; P3D_FUSED_TAIL_DIV=$1 P3D_FUSED_XT_TAIL=$2 python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e "s/.*k_fused': \([0-9.]*\).*/\1/"
  done
done
