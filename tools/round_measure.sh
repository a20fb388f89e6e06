#!/bin/bash
# Round measurement on the GPU box (from the repo root): parity tests, bench line, rocprofv3 kernel stats, PMC passes,
# phase profile.  Everything lands under gpurun_out/; the summaries to be judged are copied to profiles/ by hand.
set -e
REPO=$(pwd)
mkdir -p gpurun_out/pmc gpurun_out/stats
python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; cat gpurun_out/bench_n1.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/stats -o stats -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/stats/bench.log 2>&1)
tail -1 gpurun_out/stats/bench.log
bash tools/pmc_collect.sh > gpurun_out/pmc.log 2>&1; tail -30 gpurun_out/pmc.log
AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so python tools/phase_profile.py > gpurun_out/phase_quadruped.txt 2>&1
AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so python tools/phase_profile.py models/biped/biped.xml 4096 100 > gpurun_out/phase_biped.txt 2>&1
