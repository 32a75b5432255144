#!/bin/bash
# dev: A/B of builds of the C-ABI library on config c5 (32 x 256^3 fp16), per-stage times.  usage: ab_c5.sh reps lib...
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do echo -n "$(basename $L .so): "; P3D_CAPI_LIB=$PWD/$L python tools/dev/batched_time.py 2>&1 | tail -1; done
done
