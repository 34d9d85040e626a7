#!/bin/bash
# time every variant built by tools/ab_build.sh (GPU box):  bash tools/ab_run.sh [names...]
cd "$(dirname "$0")/.."
names="$@"; [ -z "$names" ] && names=$(ls gpurun_ab | sed 's/lib_\(.*\)\.so/\1/')
for v in $names; do
  SPLASH_HIP_LIBRARY=$PWD/gpurun_ab/lib_$v.so python tools/time_block.py 2>&1 | grep run_all
done
