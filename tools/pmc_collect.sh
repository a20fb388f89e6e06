#!/bin/bash
# Counter passes for the rollout kernel (run on the GPU box from the repo root): one rocprofv3 run per group,
# --pmc with --kernel-trace only.  Output: gpurun_out/pmc/<group>/..., summary gpurun_out/pmc_summary.csv
set -e
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES" \
  "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F32" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc/g$i -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $REPO/gpurun_out/pmc/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $REPO/gpurun_out/pmc/g$i.log; }
done
cd $REPO
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary.csv
cat gpurun_out/pmc_summary.csv
