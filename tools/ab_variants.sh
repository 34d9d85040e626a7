#!/bin/bash
# A/B of library variants on the GPU box: copies each variant over the product .so and runs the mid bench
for w in "$@"; do
  cp rsplash_amd/librsplash_hip_w$w.so rsplash_amd/librsplash_hip.so
  touch rsplash_amd/librsplash_hip.so
  python bench.py --rows 1024 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $w', d['value'], d['roofline']['avg_launch_ms'], d['spinup']['seconds'])"
done
