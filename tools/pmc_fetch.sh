#!/bin/bash
# FETCH_SIZE of one small bench run under a given library build (GPU box):
#   bash tools/pmc_fetch.sh <tag> [path/to/lib.so]   -> gpurun_out/prof_<tag>_fetch
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
[ -n "$2" ] && export SPLASH_HIP_LIBRARY=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py --rows 512 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $R/gpurun_out/${tag}_fetch.err || exit 1
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/prof_${tag}_fetch/*/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "run_all" in k or "snow_thr" in k:
        print("$tag", k, "calls", len(v), "FETCH_SIZE KiB mean", sum(v) / len(v))
PY
