# A/B of the whole spin-up (tools/time_block.py prints it) between the product and gpurun_ab variants, two row blocks, twice
for rep in 1 2; do
for lib in "" $(ls gpurun_ab/lib_*.so 2>/dev/null); do
  if [ -n "$lib" ]; then export SPLASH_HIP_LIBRARY=$PWD/$lib; else unset SPLASH_HIP_LIBRARY; fi
  ROWS=1024 ROW0=0 python tools/time_block.py 2>&1 | grep run_all
  ROWS=1024 ROW0=2048 python tools/time_block.py 2>&1 | grep run_all
done; done
