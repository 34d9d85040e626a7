#!/bin/bash
# Call-site profile of k_run_all (run on the GPU box): builds the library with -DSPL_SITES into a
# scratch copy, runs one small bench step with it and prints, per instrumented source line of
# rsplash_amd/csrc/splash_device.hpp, how many lanes and how many wavefronts executed it.
#   gpurun -- 'bash tools/site_profile.sh > gpurun_out/sites.txt'
set -e
cd "$(dirname "$0")/.."
cp rsplash_amd/librsplash_hip.so /tmp/librsplash_hip.product.so
(cd rsplash_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function \
    -DSPL_SITES -shared -o ../librsplash_hip.so splash_kernels.hip splash_statics.hip splash_capi.hip splash_grid_host.hip)
python bench.py --rows 512 --steps 1 --warmup 0 --cpu-seconds 0 2>&1 >/dev/null | grep SITE | tail -40 || true
cp /tmp/librsplash_hip.product.so rsplash_amd/librsplash_hip.so
