#!/bin/bash
# Call-site profile of k_run_all (run on the GPU box): builds the library with -DSPL_SITES into a
# scratch copy, runs one small bench step with it and prints, per instrumented source line of
# rsplash_amd/csrc/splash_device.hpp, how many lanes and how many wavefronts executed it.
#   gpurun -- 'bash tools/site_profile.sh > gpurun_out/sites.txt'
set -e
cd "$(dirname "$0")/.."
# (the Makefile's own source list: `make variant`; the product library is not touched)
make -s -C rsplash_amd/csrc variant NAME=sites EXTRA=-DSPL_SITES
SPLASH_HIP_LIBRARY=$PWD/gpurun_ab/lib_sites.so python bench.py --rows 512 --steps 1 --warmup 0 --cpu-seconds 0 2>&1 >/dev/null | grep SITE | tail -40 || true
