#!/bin/bash
# Round 6 A/B on the GPU box: for every library variant (product + gpurun_ab/lib_*.so) the tail with/without the
# wave-uniform variant, and k_run_all on two row blocks.   bash tools/round6_ab.sh > gpurun_out/r06_ab.txt
cd "$(dirname "$0")/.."
for lib in "" $(ls gpurun_ab/lib_*.so 2>/dev/null); do
  if [ -n "$lib" ]; then export SPLASH_HIP_LIBRARY=$PWD/$lib; else unset SPLASH_HIP_LIBRARY; fi
  timeout -k 10 300 python tools/tail_uniform_ab.py 1024 1536 || exit 1
  ROWS=512 ROW0=0 timeout -k 10 200 python tools/time_block.py || exit 1
  ROWS=512 ROW0=2048 timeout -k 10 200 python tools/time_block.py || exit 1
done
