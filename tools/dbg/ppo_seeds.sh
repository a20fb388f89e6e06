# final reward of the swing-up example over seeds, for the shipped library and for a mid-round build (rounding-level differences
# in the physics): the outcome of one seed is chaotic, the distribution is what can be compared
for lib in ambersim_amd/csrc/libamx.so ambersim_amd/csrc/ab/libamx_eqw.so; do
  for seed in 0 1 2 3 4 5; do
    AMX_LIBRARY=$lib timeout -k 10 200 python examples/rl/pendulum/ex_swingup.py 100000 $seed > gpurun_out/ppo_s.log 2>&1
    echo "$(basename $lib) seed $seed: $(grep 'Steps: 8028160' gpurun_out/ppo_s.log) $(grep 'Time to train' gpurun_out/ppo_s.log)"
  done
done
