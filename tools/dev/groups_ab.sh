#!/bin/bash
# dev: interleaved slab layout (P3D_FUSED_GROUPS=1) against the plain one, k_fused us.  usage: groups_ab.sh <outdir-tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { # shape groups xt mid tail
  r=$(P3D_FUSED_GROUPS=$2 P3D_FUSED_XT=$3 P3D_FUSED_XT_MID=$4 P3D_FUSED_XT_TAIL=$5 SHAPE=$1 python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e "s/.*V \([0-9]*\) F \([0-9]*\).*k_fused.: \([0-9.]*\).*/V \1 F \2 k_fused \3/")
  echo "$1 groups $2 xt $3 mid $4 tail $5 : $r"
}
( for rep in 1 2; do
  for s in 512,512,512 1024,512,512; do
    run $s 0 -1 -1 -1
    run $s 1 -1 -1 -1
    run $s 1 12 6 3
    run $s 1 10 5 2
    run $s 1 14 6 3
    run $s 1 8 4 2
    run $s 1 12 4 2
    run $s 1 16 8 4
  done
done ) 2>&1 | tee $O/groups.txt
