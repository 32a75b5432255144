#!/bin/bash
# dev: three-level slab taper sweep (big / mid / short) of the fixed-slab launch.  usage: taper_sweep.sh <outdir-tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { # xt nbig xtmid nmid xttail
  r=$(P3D_FUSED_XT=$1 P3D_FUSED_NBIG=$2 P3D_FUSED_XT_MID=$3 P3D_FUSED_NMID=$4 P3D_FUSED_XT_TAIL=$5 python tools/dev/fused_time.py 2>&1 | grep -o "'k_fused': [0-9.]*")
  echo "xt $1 nbig $2 | mid $3 x $4 | tail $5 : $r"
}
( for rep in 1 2; do
  run 11 36 5 0 2      # today's
  run 11 36 5 12 2
  run 11 36 5 16 2
  run 11 30 5 20 2
  run 11 32 6 14 2
  run 11 36 4 15 2
  run 12 32 6 12 2
  run 12 32 5 16 2
  run 12 30 6 16 3
  run 14 26 6 14 2
  run 14 24 7 14 3
  run 16 20 8 14 3
  run 16 22 6 16 2
  run 11 40 5 8 2
  run 11 36 5 20 1
  run 10 36 5 18 2
  run 12 34 4 14 2
done ) 2>&1 | tee $O/taper.txt
