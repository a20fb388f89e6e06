#!/bin/bash
# Runs the given step names on the GPU box in order; a step that hits its time limit (exit 124 / 137) ends the run - no
# further GPU step is started after a kill.  Outputs land in gpurun_out/.
#   bash tools/gpu_steps.sh tests parity bench configs stats pmc phase c4 ppo
REPO=$(pwd)
mkdir -p gpurun_out
run() {  # run <limit seconds> <log> <command...>
  local limit=$1 log=$2; shift 2
  echo "== $* (limit ${limit}s) -> $log"
  timeout -k 10 "$limit" "$@" > "$log" 2>&1
  local rc=$?
  tail -4 "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its limit (rc=$rc): stopping"; exit $rc; fi
  return 0
}
for step in "$@"; do
  case $step in
    tests)   run 1000 gpurun_out/r02_pytest_gpu.log python -m pytest tests -m gpu -q -s -x ;;
    testsall) run 1000 gpurun_out/r02_pytest_gpu.log python -m pytest tests -m gpu -q -s ;;
    parity)  run 600 gpurun_out/r02_parity.log python tools/parity_horizons.py --out gpurun_out/r02_parity_horizons.txt ;;
    bench)   run 600 gpurun_out/r02_bench_n1.log python bench.py ;;
    configs) run 600 gpurun_out/r02_configs_throughput.txt python tools/configs_bench.py ;;
    stats)   (cd /tmp && export TMPDIR=/tmp && run 600 $REPO/gpurun_out/r02_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r02_stats -o stats -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary)
             (cd /tmp && export TMPDIR=/tmp && run 600 $REPO/gpurun_out/r02_stats_biped.log rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r02_stats_biped -o stats -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --model models/biped/biped.xml) ;;
    pmc)     mkdir -p gpurun_out/pmc_quadruped gpurun_out/pmc_biped
             run 900 gpurun_out/r02_pmc_quadruped.log bash tools/pmc_collect.sh quadruped
             run 900 gpurun_out/r02_pmc_biped.log bash tools/pmc_collect.sh biped --model models/biped/biped.xml ;;
    phase)   AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so run 300 gpurun_out/r02_phase_quadruped.txt python tools/phase_profile.py
             AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so run 300 gpurun_out/r02_phase_biped.txt python tools/phase_profile.py models/biped/biped.xml 4096 100 ;;
    c4)      run 300 gpurun_out/r02_c4_fd_grad.txt python tools/c4_bench.py
             (cd /tmp && export TMPDIR=/tmp && run 600 $REPO/gpurun_out/r02_stats_c4.log rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r02_stats_c4 -o stats -- python3 $REPO/tools/c4_bench.py) ;;
    ppo)     run 600 gpurun_out/r02_ppo_pendulum_swingup.log python examples/rl/pendulum/ex_swingup.py ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
