#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of the library (build container, no GPU): every translation unit compiled with
# -fsanitize=address,undefined -fno-gpu-sanitize (the device code is left alone: GPU ASan is not available on this pool), then
# the host-backend tests (tests/test_host_backend.py: the product's own day step on the CPU, the host grid path, the module
# mirrors) run against that build.  Python is not an ASan build, so the runtime is preloaded.
#   bash tools/asan_host_backend.sh > profiles/rNN_asan_host_backend.txt
set -e
cd "$(dirname "$0")/.."
D=/tmp/splash_asan; rm -rf $D; mkdir -p $D
SRC=$(make -s -C rsplash_amd/csrc -pn 2>/dev/null | sed -n 's/^SRC *:= *//p' | head -1)
for f in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize \
      -fno-omit-frame-pointer -Wno-unused-function -c -o $D/${f%.hip}.o rsplash_amd/csrc/$f
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $D/librsplash_hip.so $D/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
echo "# sanitized host build: $SRC"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  SPLASH_HIP_LIBRARY=$D/librsplash_hip.so python -m pytest tests/test_host_backend.py tests/test_grid_host.py tests/test_readme_example.py -q -m "not gpu" -k "not cpp_host_mirror" 2>&1 | tail -5
