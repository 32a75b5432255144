#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
for i in 1 2 3 4; do
  for L in tg1 s1; do echo -n "$L: "; P3D_CAPI_LIB=$R/build_dev/$L.so python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done
done | tee $O/ab.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_slab.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
