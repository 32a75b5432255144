#!/bin/bash
# dev: how the vertex copy of a stack (config c5) is split between the counting launch and the face launch
for np_e in "2 1" "4 1" "4 2" "8 1" "8 2" "8 3" "2 0" "4 0"; do set -- $np_e
  echo -n "nparts=$1 early=$2: "; P3D_STACK_NPARTS=$1 P3D_STACK_EARLY=$2 python tools/dev/batched_time.py 2>&1 | tail -2 | tr '\n' ' '; echo
done
