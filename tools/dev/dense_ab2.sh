mkdir -p gpurun_out/r06c
run() { echo -n "$1 | "; env $2 P3D_CAPI_LIB=$PWD/$3 timeout 120 python tools/dev/dense_ab.py 2>&1 | tail -1; }
for i in 1 2; do
run base "X=1" primitive3d_amd/libp3dmc.so
run dense_atomic "X=1" build_dev/dense.so
run dense_atomic_uc "P3D_RING_MEM=3" build_dev/dense.so
run dense_load_uc "P3D_RING_MEM=3 P3D_DENSE_POLL=1" build_dev/dense.so
run dense_load_fg "P3D_RING_MEM=1 P3D_DENSE_POLL=1" build_dev/dense.so
run dense_load_uc_s8 "P3D_RING_MEM=3 P3D_DENSE_POLL=1" build_dev/dense_s8.so
run dense_load_normal "P3D_DENSE_POLL=1" build_dev/dense.so
done 2>&1 | tee gpurun_out/r06c/ab.txt
