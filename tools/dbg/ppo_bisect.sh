for lib in eqw rg2 pk3 nc1 imp1 bl1 vg1 cur; do
  AMX_LIBRARY=ambersim_amd/csrc/ab/libamx_$lib.so timeout -k 10 200 python examples/rl/pendulum/ex_swingup.py > gpurun_out/ppo_$lib.log 2>&1
  echo "$lib: $(grep 'Steps: 8028160' gpurun_out/ppo_$lib.log) $(grep 'Time to train' gpurun_out/ppo_$lib.log)"
done
