# A/B of the monthly instantiation's NA-day bookkeeping: product vs the variants under gpurun_ab/ (lib_prena.so: the library before the NA-day rule; lib_select.so was: selects every day vs a wave-uniform fast path -- no difference, 4.26 vs 4.27-4.29 ms)
for rep in 1 2; do
for lib in "" $(ls $PWD/gpurun_ab/lib_*.so); do
  if [ -n "$lib" ]; then export SPLASH_HIP_LIBRARY=$lib; else unset SPLASH_HIP_LIBRARY; fi
  python bench.py --config 4 --cpu-seconds 0 2>/dev/null | python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(os.environ.get('SPLASH_HIP_LIBRARY', 'product'), 'config 4: value %.4e, ms/step %.2f, run_all launch %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
