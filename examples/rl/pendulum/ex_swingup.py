"""Pendulum swing-up with PPO on the HIP physics path (counterpart of reference examples/rl/pendulum/ex_swingup.py,
same hyper-parameters; no viewer).  Run on an MI355X box: python examples/rl/pendulum/ex_swingup.py [num_timesteps [seed]]"""
import functools
import os
import sys
from datetime import datetime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from ambersim_amd.rl import ppo
from ambersim_amd.rl.helpers import make_ppo_networks
from ambersim_amd.rl.pendulum.swingup import PendulumSwingupEnv

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    num_envs = 1024
    env = PendulumSwingupEnv(num_envs=num_envs, device=dev)
    eval_env = PendulumSwingupEnv(num_envs=128, device=dev)
    network_factory = functools.partial(make_ppo_networks, policy_hidden_layer_sizes=(64,) * 3)
    train_fn = functools.partial(
        ppo.train,
        num_timesteps=int(sys.argv[1]) if len(sys.argv) > 1 else 100_000,
        num_evals=50,
        reward_scaling=0.1,
        episode_length=200,
        normalize_observations=True,
        action_repeat=1,
        unroll_length=10,
        num_minibatches=32,
        num_updates_per_batch=8,
        discounting=0.97,
        learning_rate=3e-4,
        entropy_cost=0,
        num_envs=num_envs,
        batch_size=512,
        network_factory=network_factory,
        seed=int(sys.argv[2]) if len(sys.argv) > 2 else 0,
    )
    times = [datetime.now()]

    def progress(num_steps, metrics):
        print(f"  Steps: {num_steps}, Reward: {metrics['eval/episode_reward']:.2f}", flush=True)
        times.append(datetime.now())

    print("Training...")
    result = train_fn(environment=env, eval_env=eval_env, progress_fn=progress)
    print(f"Time to train: {times[-1] - times[0]}")
    result.save("gpurun_out/pendulum_swingup_ppo.pt" if os.path.isdir("gpurun_out") else "pendulum_swingup_ppo.pt")
