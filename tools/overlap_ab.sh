#!/bin/bash
# A/B of library variants (tools/ab_build.sh) on bench.py's whole-call figure: the spin-up's tail overlapped with the
# transient run (GPU box):  bash tools/overlap_ab.sh [reps] name1 name2 ...
cd "$(dirname "$0")/.."
reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    SPLASH_HIP_LIBRARY=$PWD/gpurun_ab/lib_$v.so python bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', 'inclusive %.3f tail %.3f spin-up alone %.3f run_all %.2f ms' % (d['inclusive']['seconds'],d['inclusive']['tail_seconds'],d['spinup']['seconds'], 49.0/d['roofline']['achieved']*1e3*1.0))"
  done
done
