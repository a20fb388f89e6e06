#!/bin/bash
# r05: hard-impact parity and throughput of the elimination variants (tools/devbuild.sh libraries under ambersim_amd/csrc/ab/)
#   bash tools/r05_elim_ab.sh new old sq tr
mkdir -p gpurun_out
OUT=gpurun_out/r05_elim_ab.txt
: > $OUT
for t in "$@"; do
  echo "== $t" >> $OUT
  AMX_LIBRARY=ambersim_amd/csrc/ab/libamx_$t.so timeout -k 10 300 python tools/dbg/standing_impact.py >> $OUT 2>&1 || { echo "step failed/killed: stopping" >> $OUT; exit 1; }
done
LIBS=""; for t in "$@"; do LIBS="$LIBS ambersim_amd/csrc/ab/libamx_$t.so"; done
AB_CASES=biped_stand,biped,biped_rest,quadruped timeout -k 10 600 python tools/ab_bench.py $LIBS $LIBS >> $OUT 2>&1
tail -30 $OUT
