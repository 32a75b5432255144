#!/bin/bash
for n in 512 1024; do
for cfg in "8 2 4" "12 3 4" "12 4 4" "12 3 3" "14 4 4" "16 4 3" "16 4 4" "12 2 4" "10 3 4"; do
  set -- $cfg
  echo -n "N=$n XT=$1 TAIL=$2 DIV=$3: "; N=$n P3D_FUSED_XT=$1 P3D_FUSED_XT_TAIL=$2 P3D_FUSED_TAIL_DIV=$3 python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*k_fused/k_fused/'
done; done
