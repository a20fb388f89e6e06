#!/bin/bash
# LDS bank-conflict cycles of the headline model at 1 / 3 / 5 Newton iterations: what part of them is per solver iteration.  Make the variants first:
#   for it in 1 5; do sed "s/iterations=\"3\"/iterations=\"$it\"/" ambersim_amd/models/biped/biped_stand.xml > ambersim_amd/models/biped/_tmp_stand_it$it.xml; done
# and PREBUILD their kernels where hipcc runs outside the profiler (python -m ambersim_amd.prebuild ...): a build under rocprofv3 --pmc starts
# helper programs from a GPU-initialised process, which the GPU box refuses.
REPO=$(pwd); export TMPDIR=/tmp
cd /tmp
for m in _tmp_stand_it1 biped_stand _tmp_stand_it5; do
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/ldsc_$m -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --model models/biped/$m.xml > $REPO/gpurun_out/ldsc_$m.log 2>&1 || exit $?
  (cd $REPO && echo "== $m" && python3 tools/pmc_summary.py gpurun_out/ldsc_$m && tail -1 gpurun_out/ldsc_$m.log | cut -c1-100)
done
