#!/bin/bash
# Round-3 measurement on the GPU box (from the repo root): bench line, rocprofv3 kernel stats of the same command, PMC passes for
# both headline models, phase profiles (two waves per SIMD and one).  Everything lands under gpurun_out/r03m/.
REPO=$(pwd)
OUT=gpurun_out/r03m
mkdir -p $OUT/stats
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 1500 $OUT/bench_n1.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/stats -o stats -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $REPO/$OUT/stats/bench.log 2>&1)
tail -1 $OUT/stats/bench.log | cut -c1-400
bash tools/pmc_collect.sh biped > $OUT/pmc_biped.log 2>&1; tail -40 $OUT/pmc_biped.log
bash tools/pmc_collect.sh quadruped --model models/quadruped/quadruped.xml > $OUT/pmc_quadruped.log 2>&1; tail -40 $OUT/pmc_quadruped.log
AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so python tools/phase_profile.py models/biped/biped.xml 4096 1000 > $OUT/phase_biped.txt 2>&1
AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so python tools/phase_profile.py models/biped/biped.xml 1024 100 > $OUT/phase_biped_1w.txt 2>&1
AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so python tools/phase_profile.py models/quadruped/quadruped.xml 4096 1000 > $OUT/phase_quadruped.txt 2>&1
grep "cycles per env-step" $OUT/phase_*.txt
