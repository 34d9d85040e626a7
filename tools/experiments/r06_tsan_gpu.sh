# ThreadSanitizer over the device-path host code (lanes, delivery threads, pool, workers) ON the GPU box: the HIP runtime is
# not instrumented, so only reports with a frame in librsplash_hip.so and none inside libamdhip64/libhsa are of interest
set -e
D=/tmp/splash_tsan; rm -rf $D; mkdir -p $D
SRC=$(make -s -C rsplash_amd/csrc -pn 2>/dev/null | sed -n 's/^SRC *:= *//p' | head -1)
for f in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer \
      -Wno-unused-function -c -o $D/${f%.hip}.o rsplash_amd/csrc/$f &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -fno-gpu-sanitize -o $D/librsplash_hip.so $D/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
set +e
LD_PRELOAD=$RT TSAN_OPTIONS="report_signal_unsafe=0:halt_on_error=0:log_path=$D/tsan:ignore_noninstrumented_modules=1" \
  SPLASH_HIP_LIBRARY=$D/librsplash_hip.so timeout -k 10 500 python -m pytest tests/test_gpu_host_path.py -q -m gpu -x \
  -k "pipeline_lanes or cpp_exception or out_of_memory" 2>&1 | tail -4
echo "all reports: $(cat $D/tsan.* 2>/dev/null | grep -c 'WARNING: ThreadSanitizer')"
cat $D/tsan.* 2>/dev/null | awk 'BEGIN{RS="=================="} /librsplash_hip/ {n++; if (n<=4) print} END{print "reports with a frame in the library:", n+0}' | cut -c1-220 | head -120
