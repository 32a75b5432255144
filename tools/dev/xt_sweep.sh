#!/bin/bash
# dev: sweep of the streaming kernel's planes-per-block (x-halo overhead vs launch tail) on the 512^3 grid
for cfg in "8 2" "8 8" "11 11" "13 13" "16 16" "16 4" "22 22" "23 23" "24 24" "26 26" "32 32" "32 8"; do
  set -- $cfg
  echo -n "XT=$1 TAIL=$2: "; P3D_FUSED_XT=$1 P3D_FUSED_XT_TAIL=$2 python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'
done
