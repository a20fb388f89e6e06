#!/bin/bash
# Round-5 measurement on the GPU box (from the repo root): bench line, rocprofv3 kernel stats of the same command, PMC passes for
# the headline (biped_stand), phase profiles (two waves per SIMD and one).  Everything lands under gpurun_out/r05m/.
#   bash tools/r05_measure.sh [bench] [stats] [pmc] [phase]      (default: all)
REPO=$(pwd)
OUT=gpurun_out/r05m
mkdir -p $OUT/stats
STEPS=${@:-bench stats pmc phase}
for s in $STEPS; do
case $s in
bench) timeout -k 10 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 400 $OUT/bench_n1.json ;;
stats) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/stats -o stats -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $REPO/$OUT/stats/bench.log 2>&1)
       tail -1 $OUT/stats/bench.log | cut -c1-300 ;;
pmc)   bash tools/pmc_collect.sh biped_stand > $OUT/pmc_biped_stand.log 2>&1; tail -45 $OUT/pmc_biped_stand.log ;;
pmc_rest) bash tools/pmc_collect.sh biped_rest --model models/biped/biped_rest.xml > $OUT/pmc_biped_rest.log 2>&1; tail -45 $OUT/pmc_biped_rest.log ;;
phase) AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so timeout -k 10 300 python tools/phase_profile.py models/biped/biped_stand.xml 4096 1000 > $OUT/phase_biped_stand.txt 2>&1
       AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so timeout -k 10 300 python tools/phase_profile.py models/biped/biped_stand.xml 1024 100 > $OUT/phase_biped_stand_1w.txt 2>&1
       AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so timeout -k 10 300 python tools/phase_profile.py models/biped/biped_rest.xml 4096 200 > $OUT/phase_biped_rest.txt 2>&1
       grep "cycles per env-step" $OUT/phase_*.txt ;;
esac
done
# extras: bash tools/r05_measure.sh extras
for s in $STEPS; do
case $s in
extras) timeout -k 10 400 python tools/parity_horizons.py --out $OUT/parity_horizons.txt > $OUT/parity.log 2>&1; tail -14 $OUT/parity_horizons.txt
        timeout -k 10 300 python tools/variants_check.py --out $OUT/biped_variants.txt > $OUT/variants.log 2>&1; cat $OUT/biped_variants.txt
        timeout -k 10 200 python tools/c4_bench.py > $OUT/c4_fd_grad.txt 2>&1; AMX_NO_WIDE=1 timeout -k 10 200 python tools/c4_bench.py >> $OUT/c4_fd_grad.txt 2>&1; cat $OUT/c4_fd_grad.txt
        timeout -k 10 300 python tools/ppo_breakdown.py > $OUT/ppo_breakdown.txt 2>&1; tail -14 $OUT/ppo_breakdown.txt
        timeout -k 10 400 python examples/rl/pendulum/ex_swingup.py 8028160 > $OUT/ppo_pendulum_swingup.log 2>&1; tail -3 $OUT/ppo_pendulum_swingup.log ;;
esac
done
