#!/bin/bash
# build kernel variants for A/B timing:  tools/ab_build.sh name1 "flags1" name2 "flags2" ...
# -> gpurun_ab/lib_<name>.so (travels to the GPU box; the product .so is not touched)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_ab
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  (cd rsplash_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags \
      -shared -o ../../gpurun_ab/lib_$name.so splash_kernels.hip splash_statics.hip splash_capi.hip splash_grid_host.hip) &
done
wait
ls -la gpurun_ab
