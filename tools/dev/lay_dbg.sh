for L in primitive3d_amd/libp3dmc.so build_dev/laydbg64.so; do echo "== $L"; P3D_CAPI_LIB=$PWD/$L REPS=1 python tools/dev/layout_time.py 2>&1 | tail -1 | cut -c1-200; done
