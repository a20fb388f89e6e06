#!/bin/bash
# Fast A/B builds of the kernels: the shipped specialisations only (no generic kernels: a third of the compile time).
#   bash tools/devbuild.sh <tag> [extra hipcc flags...]   -> ambersim_amd/csrc/ab/libamx_<tag>.so (+ .s with ASM=1)
# tools/ab_bench.py runs such libraries side by side on the GPU box (AMX_LIBRARY).
set -e
TAG=$1; shift
cd "$(dirname "$0")/../ambersim_amd/csrc"
mkdir -p ab
make -s amx_specs_gen.h
# SPECS="biped,biped_stand": only those records (a third of the compile time again)
if [ -n "$SPECS" ]; then
  python3 ../../tools/gen_specs.py --only "$SPECS" --out "$PWD/ab/amx_specs_$TAG.h"
  set -- "$@" "-DAMX_SPECS_HEADER=\"$PWD/ab/amx_specs_$TAG.h\""
fi
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -DAMX_NO_GENERIC=1"
if [ -n "$ASM" ]; then
  /opt/rocm/bin/hipcc $FLAGS "$@" --cuda-device-only -S -o ab/amx_$TAG.s -x hip amx_api.cpp -Rpass-analysis=kernel-resource-usage 2> ab/amx_$TAG.remarks || { tail -20 ab/amx_$TAG.remarks; exit 1; }
fi
/opt/rocm/bin/hipcc $FLAGS "$@" -no-hip-rt -shared -o ab/libamx_$TAG.so -x hip amx_api.cpp
ls -la ab/libamx_$TAG.so
