#!/bin/bash
# ThreadSanitizer over the HOST side of the library (build container, no GPU): every translation unit compiled with
# -fsanitize=thread -fno-gpu-sanitize, then the tests that start host threads without a device -- the host backend's
# cell pool (splash_grid_run with SPLASH_BACKEND_HOST), splash_grid_submit / _wait job threads, the raster-level loop with
# three blocks in flight, the fault-injection throw on the job thread -- run against that build.  Python is not a TSan
# build, so the runtime is preloaded; races inside the interpreter or numpy are not ours and are filtered by module.
#   bash tools/tsan_host_backend.sh > profiles/rNN_tsan_host_backend.txt
set -e
cd "$(dirname "$0")/.."
D=/tmp/splash_tsan; rm -rf $D; mkdir -p $D
SRC=$(make -s -C rsplash_amd/csrc -pn 2>/dev/null | sed -n 's/^SRC *:= *//p' | head -1)
for f in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-gpu-sanitize \
      -fno-omit-frame-pointer -Wno-unused-function -c -o $D/${f%.hip}.o rsplash_amd/csrc/$f &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -fno-gpu-sanitize -o $D/librsplash_hip.so $D/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
echo "# thread-sanitized host build: $SRC"
set +e
LD_PRELOAD=$RT TSAN_OPTIONS="report_signal_unsafe=0:halt_on_error=0:second_deadlock_stack=1:log_path=$D/tsan" \
  SPLASH_HIP_LIBRARY=$D/librsplash_hip.so python -m pytest tests/test_host_backend.py tests/test_grid_host.py -q -m "not gpu" \
  -k "grid or exception or chunks" 2>&1 | tail -4
echo "# reports that name the library (librsplash_hip.so):"
cat $D/tsan.* 2>/dev/null | grep -c "WARNING: ThreadSanitizer" | sed 's/^/#   all reports: /'
n=0
for f in $D/tsan.*; do [ -f "$f" ] || continue; awk 'BEGIN{RS="=================="} /librsplash_hip/ {print; n++} END{exit n==0}' "$f" && n=$((n+1)); done
echo "#   reports with a frame in the library: $n file(s)"
