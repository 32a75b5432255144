#!/bin/bash
# dev: A/B of the middle slab level (P3D_FUSED_NMID=0: big + short slabs only) on several shapes, k_fused us.  usage: mid_ab.sh <outdir-tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
( for s in 512,512,512 500,500,500 768,768,768 384,512,512 513,511,517 300,700,450 1024,1024,1024 512,512,2048 256,1024,512 640,640,640; do
    for rep in 1 2; do for m in 0 -1; do
      echo -n "SHAPE=$s nmid=$m: "; P3D_FUSED_NMID=$m SHAPE=$s python tools/dev/fused_time.py 2>&1 | tail -1 | grep -o "'k_fused': [0-9.]*"
    done; done
  done ) 2>&1 | tee $O/mid_ab.txt
