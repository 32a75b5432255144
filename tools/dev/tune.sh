#!/bin/bash
# usage: tools/dev/tune.sh  -- sweeps the fused kernel's launch geometry on the 512^3 bench
for xt in 8 16 32 64; do
  echo "== XT=$xt"; P3D_FUSED_XT=$xt python bench.py --steps 10 --warmup 3 --stages --no-cpu-baseline 2>&1 | grep -E "stage ms|value" | sed -e 's/"config".*"roofline"/"roofline"/' | cut -c1-420
done
