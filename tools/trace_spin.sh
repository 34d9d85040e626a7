#!/bin/bash
# kernel trace of a small bench run; prints the spin-up launches in order (GPU box)
tag=$1; rows=${2:-1024}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
[ -n "$3" ] && export SPLASH_HIP_LIBRARY=$3
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_spin -- python3 $R/bench.py --rows $rows --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $R/gpurun_out/${tag}_spin.err || exit 1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/prof_${tag}_spin/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
n = 0
for r in rows:
    k = r["Kernel_Name"]
    if "spin_phase" not in k:
        continue
    if t0 is None:
        t0 = int(r["Start_Timestamp"])
    n += 1
    if n > 40:
        break
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} ms  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  {k[:48]}")
PY
