#!/bin/bash
# dev: device assembly of p3d_mc.hip (+ resource usage) into build_dev/isa/<tag>.s   usage: tools/dev/isa.sh <tag> [-Dflags...]
tag=$1; shift
mkdir -p build_dev/isa/$tag && cd build_dev/isa/$tag && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None --save-temps \
  -Rpass-analysis=kernel-resource-usage "$@" ../../../primitive3d_amd/csrc/p3d_mc.hip -o lib.so 2> rpass.txt
mv p3d_mc-hip-amdgcn-amd-amdhsa-gfx950.s ../$tag.s && rm -f p3d_mc-h* p3d_mc.hip-hip*
echo build_dev/isa/$tag.s
