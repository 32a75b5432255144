#!/bin/bash
# dev: an alternative build of the C-ABI library for A/B runs (P3D_CAPI_LIB=build_dev/<name>.so).  usage: build_variant.sh <name> [-D...]
name=$1; shift
mkdir -p build_dev
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" primitive3d_amd/csrc/p3d_mc.hip -o build_dev/$name.so && echo build_dev/$name.so
