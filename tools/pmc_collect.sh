#!/bin/bash
# Counter passes for the rollout kernel (run on the GPU box from the repo root): one rocprofv3 run per group,
# --pmc with --kernel-trace only (never combined with other trace domains).  Usage: bash tools/pmc_collect.sh <tag> [bench.py args]
# Output: gpurun_out/pmc_<tag>/g<i>/..., summary gpurun_out/pmc_<tag>_summary.csv
TAG=${1:-biped}; shift  # bench.py's default workload is the biped; quadruped: bash tools/pmc_collect.sh quadruped --model models/quadruped/quadruped.xml
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out/pmc_$TAG
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES" \
  "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F32" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$TAG/g$i -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $REPO/gpurun_out/pmc_$TAG/g$i.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "group $i killed at its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && { echo "group $i failed (rc=$rc)"; tail -3 $REPO/gpurun_out/pmc_$TAG/g$i.log; }
done
cd $REPO
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_${TAG}_summary.csv
cat gpurun_out/pmc_${TAG}_summary.csv
