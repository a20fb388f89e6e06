#!/bin/bash
# r05: distribution of the one-step hard-impact error over NENV environments per library variant
mkdir -p gpurun_out
OUT=gpurun_out/r05_impact_stats.txt
: > $OUT
for t in "$@"; do
  echo "== $t" >> $OUT
  NENV=${NENV:-512} AMX_LIBRARY=ambersim_amd/csrc/ab/libamx_$t.so timeout -k 10 400 python tools/dbg/standing_impact.py 2>&1 | grep -v amdgpu.ids >> $OUT || { echo "step failed/killed: stopping" >> $OUT; exit 1; }
done
cat $OUT
