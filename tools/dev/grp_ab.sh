# dev: the cursor-group rule for even column counts, old against new, on the shapes it changes (same box, interleaved)
mkdir -p gpurun_out/r06h
for i in 1 2 3; do for S in 1024,1024,1024 1024,1024,512 129,1024,1024; do for L in primitive3d_amd/libp3dmc.so build_dev/oldgrp.so; do
  echo -n "$S $(basename $L) | "; SHAPE=$S P3D_CAPI_LIB=$PWD/$L timeout 200 python tools/dev/dense_ab.py 2>&1 | tail -1 | cut -c1-220
done; done; done | tee gpurun_out/r06h/grp_ab.txt
