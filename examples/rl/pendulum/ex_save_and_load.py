"""Swing-up training with custom policy/value architectures, then saving and reloading the trained networks (counterpart of
reference examples/rl/pendulum/ex_save_and_load.py; no viewer).  Run on an MI355X box:
python examples/rl/pendulum/ex_save_and_load.py"""
import functools
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from ambersim_amd.learning.architectures import MLP
from ambersim_amd.rl import ppo
from ambersim_amd.rl.helpers import NormalTanhDistribution, PPONetworksWrapper
from ambersim_amd.rl.pendulum.swingup import PendulumSwingupEnv

PARAMS_PATH, NETWORKS_PATH = "/tmp/pendulum_params.pt", "/tmp/pendulum_networks.pkl"


def train_swingup(dev):
    env = PendulumSwingupEnv(num_envs=1024, device=dev)
    network_wrapper = PPONetworksWrapper(
        policy_network=MLP(layer_sizes=(512, 4, 128, 2)),  # outputs mean and scale
        value_network=MLP(layer_sizes=(26, 128, 53, 1)),  # outputs a single value
        action_distribution=NormalTanhDistribution,
    )
    result = ppo.train(environment=env, eval_env=PendulumSwingupEnv(num_envs=128, device=dev), num_timesteps=100_000, num_evals=10,
                       reward_scaling=0.1, episode_length=200, normalize_observations=True, action_repeat=1, unroll_length=10,
                       num_minibatches=32, num_updates_per_batch=8, discounting=0.97, learning_rate=3e-4, entropy_cost=0, num_envs=1024,
                       batch_size=512, network_factory=network_wrapper.make_ppo_networks, seed=0,
                       progress_fn=lambda n, m: print(f"  Steps: {n}, Reward: {m['eval/episode_reward']:.2f}", flush=True))
    print("Saving...")
    result.save(PARAMS_PATH)
    with open(NETWORKS_PATH, "wb") as f:
        pickle.dump(network_wrapper, f)
    return result


def test_trained_swingup_policy(dev, trained):
    print("Loading trained policy...")
    with open(NETWORKS_PATH, "rb") as f:
        network_wrapper = pickle.load(f)
    env = PendulumSwingupEnv(num_envs=128, device=dev)
    nets = network_wrapper.make_ppo_networks(observation_size=env.observation_size, action_size=env.action_size)
    nets.policy_network.to(dev); nets.value_network.to(dev)
    loaded = ppo.PPOResult(nets, ppo.RunningStatistics(env.observation_size, dev), [])
    loaded.load(PARAMS_PATH)
    r0 = ppo.evaluate(env, trained.make_inference_fn(), 200, seed=5)
    r1 = ppo.evaluate(env, loaded.make_inference_fn(), 200, seed=5)
    print(f"episode reward: trained {r0:.2f}, reloaded {r1:.2f}")
    assert abs(r0 - r1) < 1e-3 * max(1.0, abs(r0))


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    test_trained_swingup_policy(dev, train_swingup(dev))
