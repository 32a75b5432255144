#!/bin/bash
# dev: where the middle slab level should sit (tail_div = share of big slabs, mid length, mid count), k_fused us.  usage: mid_sweep.sh <outdir-tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { # shape taildiv xtmid nmid
  r=$(P3D_FUSED_TAIL_DIV=$2 P3D_FUSED_XT_MID=$3 P3D_FUSED_NMID=$4 SHAPE=$1 python tools/dev/fused_time.py 2>&1 | grep -o "'k_fused': [0-9.]*")
  echo "$1 taildiv $2 mid $3 x $4 : $r"
}
( for rep in 1 2; do
  for s in 512,512,512 513,511,517; do
    run $s 4 -1 -1
    run $s 4 5 18
    run $s 4 5 20
    run $s 4 6 14
    run $s 4 7 12
    run $s 3 5 22
    run $s 3 6 20
    run $s 3 7 16
    run $s 5 5 12
    run $s 5 4 16
    run $s 6 5 10
    run $s 3 8 14
  done
done ) 2>&1 | tee $O/mid_sweep.txt
