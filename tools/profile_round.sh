#!/bin/bash
# rocprofv3 passes of one round (GPU box):  bash tools/profile_round.sh r02a
# kernel trace + stats on the default bench, then three PMC passes (each its own run, only combined
# with --kernel-trace) on a 1024-row bench; outputs under gpurun_out/prof_<tag>_*; condense with
# tools/summarize_profile.py.  The program itself follows `--` (never env/bash -c: see the guide).
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_trace -- python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err || exit 1
small="--rows 1024 --steps 1 --warmup 0 --cpu-seconds 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py $small > /dev/null 2> $R/gpurun_out/${tag}_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_write -- python3 $R/bench.py $small > /dev/null 2> $R/gpurun_out/${tag}_write.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_sq -- python3 $R/bench.py $small > /dev/null 2> $R/gpurun_out/${tag}_sq.err || exit 1
echo profiled $tag
