#!/bin/bash
# dev: the randomized parity campaigns + odd full-size shapes in one GPU call.  usage: fuzz_round.sh <outdir-tag> [N]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; N=${2:-150}; mkdir -p $O; cd $R
( for s in 21 22 23; do SEED=$s N=$N python tools/dev/fuzz_parity.py; done
  for s in 31 32; do SEED=$s N=$N DTYPE=f16 python tools/dev/fuzz_parity.py; done
  SEED=41 N=40 RZ=2040,2300 python tools/dev/fuzz_parity.py
  for s in 51 52; do SEED=$s N=60 python tools/dev/fuzz_batched.py; done
  for s in 61 62; do SEED=$s N=60 python tools/dev/fuzz_slabs.py; done
  python tools/dev/odd_shapes.py
  python tools/dev/odd_fields.py ) 2>&1 | grep -v "amdgpu.ids" | tee $O/fuzz.txt | tail -40
