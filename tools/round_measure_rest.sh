#!/bin/bash
# The second half of tools/round_measure.sh (everything but the rocprofv3 passes), for GPU calls with a 20-minute limit:
#   bash tools/round_measure_rest.sh r06
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for c in 1 3 4; do
  python bench.py --config $c --cpu-seconds 0 > gpurun_out/${tag}_cfg$c.json 2> gpurun_out/${tag}_cfg$c.err || echo "config $c failed"
  echo "config $c done"
done
SPLASH_BENCH_BACKEND=gloo SPLASH_BENCH_ONE_GPU=1 python bench.py --gpus 2 --rows 256 --block-rows 256 --cpu-seconds 0 > gpurun_out/${tag}_n2.json 2> gpurun_out/${tag}_n2.err || echo "n2 failed"
echo "n2 done"
( python tools/bench_host_path.py 1024 4096 daily 5 f32; python tools/bench_host_path.py 1024 4096 daily 5 f64; python tools/bench_host_path.py 1024 4096 monthly 9 f32 ) > gpurun_out/${tag}_host_path.txt 2>&1
echo "host path done"
python tools/bench_unswc.py > gpurun_out/${tag}_unswc.json 2> gpurun_out/${tag}_unswc.err
python tools/bench_terrain.py > gpurun_out/${tag}_terrain.json 2> gpurun_out/${tag}_terrain.err
python tools/bench_k2.py > gpurun_out/${tag}_k2.txt 2>&1
echo "all done"
