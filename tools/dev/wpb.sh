for L in libp3dmc_wpb8.so libp3dmc_wpb6.so; do P3D_CAPI_LIB=$PWD/primitive3d_amd/$L python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1; done
for i in 1 2 3; do for L in libp3dmc.so libp3dmc_wpb8.so libp3dmc_wpb6.so; do echo -n "$L: "; P3D_CAPI_LIB=$PWD/primitive3d_amd/$L python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done; done
for L in libp3dmc.so libp3dmc_wpb8.so; do echo -n "1024 $L: "; N=1024 P3D_CAPI_LIB=$PWD/primitive3d_amd/$L python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done
