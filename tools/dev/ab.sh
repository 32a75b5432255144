#!/bin/bash
# dev: A/B two builds of the C-ABI library on the same box.  usage: tools/dev/ab.sh libA.so libB.so [reps]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    echo -n "$(basename $L): "; P3D_CAPI_LIB=$PWD/$L python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'
  done
done
