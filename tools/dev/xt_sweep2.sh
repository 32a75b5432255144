#!/bin/bash
# dev: second sweep of the streaming kernel's slab shape (planes per block / tail planes / share of tapered slabs)
for cfg in "8 2 4" "8 1 4" "8 4 4" "8 2 8" "8 2 3" "8 2 2" "10 2 4" "12 3 4" "6 2 4" "8 3 4" "16 2 4" "16 4 3"; do
  set -- $cfg
  echo -n "XT=$1 TAIL=$2 DIV=$3: "; P3D_FUSED_XT=$1 P3D_FUSED_XT_TAIL=$2 P3D_FUSED_TAIL_DIV=$3 python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*k_fused/k_fused/'
done
