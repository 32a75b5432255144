#!/bin/bash
# dev: planes-per-block / taper sweep of the streaming kernel (P3D_CAPI_LIB from the environment)
for cfg in "12 3 4" "12 12 4" "16 4 4" "16 4 8" "16 16 4" "20 5 4" "24 6 4" "24 6 8" "24 24 4" "32 8 4" "32 8 8" "12 3 8" "12 6 4" "16 8 4"; do
  set -- $cfg
  echo -n "XT=$1 TAIL=$2 DIV=$3: "; P3D_FUSED_XT=$1 P3D_FUSED_XT_TAIL=$2 P3D_FUSED_TAIL_DIV=$3 python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'
done
