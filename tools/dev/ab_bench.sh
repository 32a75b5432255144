#!/bin/bash
# dev: A/B two builds in the back-to-back call stream of bench.py (the pybind module links libp3dmc.so by rpath, so
# the library file itself is swapped).  usage: tools/dev/ab_bench.sh libA.so libB.so [reps]
A=$1; B=$2; N=${3:-3}
cp primitive3d_amd/libp3dmc.so /tmp/libp3dmc.keep
for i in $(seq $N); do
  for L in $A $B; do
    cp $L primitive3d_amd/libp3dmc.so
    echo -n "$(basename $L): "; python bench.py --no-cpu-baseline --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
  done
done
cp /tmp/libp3dmc.keep primitive3d_amd/libp3dmc.so
