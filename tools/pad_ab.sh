#!/bin/bash
# A/B on one box: bench.py's whole-call figure with and without the LDS pad of the tail's spread launches
# (SPLASH_HIP_TAIL_LDS_PAD=0 switches it off):  bash tools/pad_ab.sh [reps]
cd "$(dirname "$0")/.."
for r in $(seq ${1:-5}); do
  for pad in 0 57344; do
    SPLASH_HIP_TAIL_LDS_PAD=$pad python bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('pad $pad:', 'inclusive %.3f tail %.3f spin-up alone %.3f' % (d['inclusive']['seconds'],d['inclusive']['tail_seconds'],d['spinup']['seconds']))"
  done
done
