for i in 1 2; do for L in primitive3d_amd/libp3dmc.so build_dev/layk1.so; do echo "== $L"; P3D_CAPI_LIB=$PWD/$L REPS=1 python tools/dev/layout_time.py 2>&1 | tail -2 | cut -c1-190; done; done
