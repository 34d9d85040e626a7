for rep in 1 2; do
for lib in "" gpurun_ab/lib_prev.so; do
  if [ -n "$lib" ]; then export SPLASH_HIP_LIBRARY=$PWD/$lib; else unset SPLASH_HIP_LIBRARY; fi
  ROWS=512 ROW0=0 python tools/time_block.py 2>&1 | grep run_all
  ROWS=512 ROW0=2048 python tools/time_block.py 2>&1 | grep run_all
done; done
