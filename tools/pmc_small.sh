#!/bin/bash
# Wait/issue counters of k_run_all on a small grid (one wavefront per SIMD): bash tools/pmc_small.sh <tag> [config=1]
tag=$1; cfg=${2:-1}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_small -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/gpurun_out/${tag}_small.err || exit 1
python3 - <<PY
import csv, glob, collections
d = "$R/gpurun_out/prof_${tag}_small"
rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
disp = collections.defaultdict(dict)
for r in rows:
    disp[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for r in csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (i, k), c in sorted(disp.items(), key=lambda kv: int(kv[0][0])):
    if "k_run_all" not in k: continue
    wc = c.get("SQ_WAVE_CYCLES", 0)
    print(i, k[:40], f"{dur.get(i,0):.3f} ms waves {int(c.get('SQ_WAVES',0))} VALU {c.get('SQ_INSTS_VALU',0):.3g} SALU {c.get('SQ_INSTS_SALU',0):.3g}",
          f"wave-cycles(x4) {wc:.3g}; of them issuing VALU {c.get('SQ_ACTIVE_INST_VALU',0)/max(wc,1):.2f}, waiting for anything {c.get('SQ_WAIT_ANY',0)/max(wc,1):.2f}, waiting to issue {c.get('SQ_WAIT_INST_ANY',0)/max(wc,1):.2f}")
PY
