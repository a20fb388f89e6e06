#!/bin/bash
# LDS bank-conflict share of the falls-and-rests biped's rollout kernel under a diagnostic library and planner switches (one rocprofv3 --pmc pass):
#   bash tools/dbg/lds_conflicts.sh <tag> <library> [VAR=value ...]      -> gpurun_out/ldsc_<tag>.txt      (MODEL=path: another model)
TAG=$1; LIB=$2; shift 2
REPO=$(pwd)
export TMPDIR=/tmp AMX_SPECIALISE=0 AMX_LIBRARY=$REPO/$LIB
for kv in "$@"; do export "$kv"; done
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $REPO/gpurun_out/ldsc_$TAG -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --model ${MODEL:-models/biped/biped_rest.xml} > $REPO/gpurun_out/ldsc_$TAG.log 2>&1 || exit $?
cd $REPO
python3 tools/pmc_summary.py gpurun_out/ldsc_$TAG > gpurun_out/ldsc_$TAG.txt
tail -1 gpurun_out/ldsc_$TAG.log | cut -c1-120 >> gpurun_out/ldsc_$TAG.txt
cat gpurun_out/ldsc_$TAG.txt
