#!/bin/bash
# Instruction mix and cycles per instruction of the spin-up's kernels, the tail launch in particular
# (GPU box):  bash tools/pmc_tail.sh <tag> [rows]   -> gpurun_out/prof_<tag>_tail + a summary on stdout
tag=$1
rows=${2:-1024}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_tail -- python3 $R/bench.py --rows $rows --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $R/gpurun_out/${tag}_tail.err || exit 1
python3 - <<PY
import csv, glob, collections
d = "$R/gpurun_out/prof_${tag}_tail"
rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
# one record per dispatch: counters by name
disp = collections.defaultdict(dict)
for r in rows:
    disp[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
try:
    for r in csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
except Exception:
    pass
print("dispatch kernel ms waves VALU SALU SMEM LDS VMEM_RD  wave-cycles  cycles/instr  VALU-share")
for (i, k), c in sorted(disp.items(), key=lambda kv: int(kv[0][0])):
    if "spin_phase" not in k:
        continue
    n = sum(c.get(x, 0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"))
    if n == 0:
        continue
    # SQ_WAVE_CYCLES counts in units of 4 cycles on gfx9 (quad-cycles)
    print(i, k[:42], f"{dur.get(i, 0):9.3f}", int(c.get("SQ_WAVES", 0)), *(f"{c.get(x, 0):.3g}" for x in
          ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")),
          f"{c.get('SQ_WAVE_CYCLES', 0):.4g}", f"{4 * c.get('SQ_WAVE_CYCLES', 0) / n:.2f}",
          f"{c.get('SQ_INSTS_VALU', 0) / n:.2f}")
PY
