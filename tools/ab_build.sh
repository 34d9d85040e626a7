#!/bin/bash
# build kernel variants for A/B timing:  tools/ab_build.sh name1 "flags1" name2 "flags2" ...
# -> gpurun_ab/lib_<name>.so (travels to the GPU box; the product .so is not touched).  The source list is the
# Makefile's (`make variant`), so a new translation unit cannot be forgotten here.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_ab
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  make -s -C rsplash_amd/csrc variant NAME="$name" EXTRA="$flags" &
done
wait
ls -la gpurun_ab
