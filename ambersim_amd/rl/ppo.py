"""A torch PPO trainer that drives `MjxEnv.step`.

The reference trains with brax's `ppo.train` (reference examples/rl/pendulum/ex_swingup.py:25-43); brax and jax are
absent here, so this is a counterpart with the same knobs and the same batching arithmetic: every training step
collects `batch_size * num_minibatches * unroll_length` transitions from `num_envs` parallel envs, computes GAE
(lambda 0.95) on unrolls of `unroll_length`, and runs `num_updates_per_batch` epochs of `num_minibatches` clipped-
surrogate updates (clip 0.3, value coefficient 0.5) with Adam.  Observations are normalised by running statistics,
rewards scaled by `reward_scaling`, actions are tanh-squashed Gaussians.  Everything stays on the env's device; the
only physics call is `env.step` (one fused `amx_step` launch per control step for all envs)."""
import math
import time
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

from ambersim_amd.rl.helpers import PPONetworks, make_ppo_networks
from ambersim_amd.rl.wrappers import EpisodeAutoReset


class RunningStatistics:
    """Running mean / std of observations (brax `running_statistics`), clipped like brax's normaliser."""

    def __init__(self, size: int, device=None, max_abs_value: float = 5.0, min_std: float = 1e-6):
        self.count = torch.zeros((), device=device, dtype=torch.float64)
        self.mean = torch.zeros(size, device=device, dtype=torch.float64)
        self.m2 = torch.zeros(size, device=device, dtype=torch.float64)
        self.max_abs_value, self.min_std = max_abs_value, min_std

    def update(self, batch: torch.Tensor):
        b = batch.reshape(-1, batch.shape[-1]).to(torch.float64)
        n = b.shape[0]
        tot = self.count + n
        delta = b.mean(0) - self.mean
        self.m2 = self.m2 + ((b - b.mean(0)) ** 2).sum(0) + delta ** 2 * self.count * n / tot
        self.mean = self.mean + delta * n / tot
        self.count = tot

    @property
    def std(self):
        return torch.sqrt(self.m2 / torch.clamp(self.count, min=1.0)).clamp(min=self.min_std)

    def normalize(self, obs: torch.Tensor) -> torch.Tensor:
        out = (obs - self.mean.to(obs.dtype)) / self.std.to(obs.dtype)
        return out.clamp(-self.max_abs_value, self.max_abs_value)

    def state_dict(self):
        return {"count": self.count, "mean": self.mean, "m2": self.m2}

    def load_state_dict(self, d):
        self.count, self.mean, self.m2 = d["count"], d["mean"], d["m2"]


@dataclass
class PPOResult:
    networks: PPONetworks
    normalizer: Optional[RunningStatistics]
    metrics: list

    def make_inference_fn(self, deterministic: bool = True) -> Callable:
        nets, norm = self.networks, self.normalizer

        def policy(obs: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
            with torch.no_grad():
                x = norm.normalize(obs) if norm is not None else obs
                params = nets.policy_network(x)
                dist = nets.parametric_action_distribution
                return dist.mode(params) if deterministic else dist.postprocess(dist.sample_raw(params, generator))
        return policy

    def save(self, path: str):
        torch.save({"policy": self.networks.policy_network.state_dict(), "value": self.networks.value_network.state_dict(),
                    "normalizer": None if self.normalizer is None else self.normalizer.state_dict()}, path)

    def load(self, path: str):
        d = torch.load(path, map_location="cpu")
        self.networks.policy_network.load_state_dict(d["policy"])
        self.networks.value_network.load_state_dict(d["value"])
        if self.normalizer is not None and d["normalizer"] is not None:
            dev = self.normalizer.mean.device
            self.normalizer.load_state_dict({k: v.to(dev) for k, v in d["normalizer"].items()})


def evaluate(env, policy, episode_length: int, seed: int) -> float:
    """Mean undiscounted return of `policy` over one episode in every env of `env` (brax's `eval/episode_reward`)."""
    rng = torch.Generator(device=env.sys.device).manual_seed(seed)
    s = env.reset(rng)
    total = torch.zeros_like(s.reward)
    for _ in range(episode_length):
        s = env.step(s, policy(s.obs))
        total = total + s.reward
    return float(total.mean().item())


def train(environment, num_timesteps: int, episode_length: int, num_envs: int = 1, num_evals: int = 1, action_repeat: int = 1,
          unroll_length: int = 10, batch_size: int = 32, num_minibatches: int = 16, num_updates_per_batch: int = 2, learning_rate: float = 1e-4,
          entropy_cost: float = 1e-4, discounting: float = 0.9, reward_scaling: float = 1.0, normalize_observations: bool = False,
          clipping_epsilon: float = 0.3, gae_lambda: float = 0.95, seed: int = 0, network_factory: Callable = make_ppo_networks,
          progress_fn: Callable = lambda *a: None, eval_env=None, deterministic_eval: bool = True) -> PPOResult:
    """Argument names and defaults follow brax's `ppo.train` as called by the reference example.  `environment` must be
    built with `num_envs` parallel envs; `eval_env` (optional, same class) is used for the `num_evals` evaluations."""
    env = EpisodeAutoReset(environment, episode_length, action_repeat)
    assert environment.num_envs == num_envs, "build the environment with num_envs parallel envs"
    assert (batch_size * num_minibatches) % num_envs == 0, "batch_size * num_minibatches must be a multiple of num_envs"
    dev = environment.sys.device
    torch.manual_seed(seed)
    gen = torch.Generator(device=dev).manual_seed(seed)
    obs_size, act_size = environment.observation_size, environment.action_size
    nets = network_factory(obs_size, act_size)  # e.g. make_ppo_networks or PPONetworksWrapper(...).make_ppo_networks
    nets.policy_network.to(dev); nets.value_network.to(dev)
    dist = nets.parametric_action_distribution
    norm = RunningStatistics(obs_size, dev) if normalize_observations else None
    nets.policy_network(torch.zeros((1, obs_size), device=dev)); nets.value_network(torch.zeros((1, obs_size), device=dev))  # materialise lazy layers
    opt = torch.optim.Adam(list(nets.policy_network.parameters()) + list(nets.value_network.parameters()), lr=learning_rate)

    env_steps_per_training_step = batch_size * unroll_length * num_minibatches * action_repeat
    num_evals_after_init = max(num_evals - 1, 1)
    training_steps_per_epoch = max(1, math.ceil(num_timesteps / (num_evals_after_init * env_steps_per_training_step)))
    unrolls_per_step = batch_size * num_minibatches // num_envs
    result = PPOResult(nets, norm, [])

    def run_eval(steps_done, t0):
        e = eval_env if eval_env is not None else environment
        r = evaluate(e, result.make_inference_fn(deterministic_eval), episode_length // action_repeat, seed + 1)
        m = {"eval/episode_reward": r, "walltime": time.time() - t0}
        result.metrics.append((steps_done, m))
        progress_fn(steps_done, m)

    t0 = time.time()
    state = env.reset(gen)
    if num_evals > 1:
        run_eval(0, t0)
    steps_done = 0
    for _ in range(num_evals_after_init):
        for _ in range(training_steps_per_epoch):
            # ---- collect unrolls_per_step x unroll_length transitions from every env
            obs_b, raw_b, logp_b, rew_b, done_b, trunc_b, boot_b = [], [], [], [], [], [], []
            with torch.no_grad():
                for _u in range(unrolls_per_step):
                    o_l, a_l, lp_l, r_l, d_l, tr_l, nobs_l = [], [], [], [], [], [], []
                    for _t in range(unroll_length):
                        obs = state.obs
                        x = norm.normalize(obs) if norm is not None else obs
                        params = nets.policy_network(x)
                        raw = dist.sample_raw(params, gen)
                        state = env.step(state, dist.postprocess(raw))
                        o_l.append(obs); a_l.append(raw); lp_l.append(dist.log_prob(params, raw))
                        r_l.append(state.reward * reward_scaling); d_l.append(state.done); tr_l.append(state.info["truncation"])
                        nobs_l.append(state.info["terminal_obs"])
                    obs_b.append(torch.stack(o_l)); raw_b.append(torch.stack(a_l)); logp_b.append(torch.stack(lp_l))
                    rew_b.append(torch.stack(r_l)); done_b.append(torch.stack(d_l)); trunc_b.append(torch.stack(tr_l))
                    boot_b.append(torch.stack(nobs_l))
                # [T, U*N, ...]
                cat = lambda xs: torch.cat(xs, dim=1)
                obs_t, raw_t, logp_t, rew_t, done_t, trunc_t, nobs_t = map(cat, (obs_b, raw_b, logp_b, rew_b, done_b, trunc_b, boot_b))
                if norm is not None:
                    norm.update(obs_t)
                nx = lambda o: norm.normalize(o) if norm is not None else o
                values = nets.value_network(nx(obs_t)).squeeze(-1)
                next_values = nets.value_network(nx(nobs_t)).squeeze(-1)  # value of the observation each step led to
                # GAE with per-step successor values (handles auto-reset inside an unroll)
                term = done_t * (1 - trunc_t)
                deltas = rew_t + discounting * (1 - term) * next_values - values
                adv = torch.zeros_like(rew_t)
                acc = torch.zeros_like(rew_t[0])
                for t in range(unroll_length - 1, -1, -1):
                    acc = deltas[t] + discounting * gae_lambda * (1 - done_t[t]) * acc
                    adv[t] = acc
                targets = adv + values
                flat = lambda z: z.reshape((-1,) + z.shape[2:])
                obs_f, raw_f, logp_f, adv_f, tgt_f = map(flat, (nx(obs_t), raw_t, logp_t, adv, targets))
                adv_f = (adv_f - adv_f.mean()) / (adv_f.std() + 1e-8)
            steps_done += env_steps_per_training_step
            # ---- updates: minibatches of batch_size * unroll_length transitions
            n = obs_f.shape[0]
            mb = n // num_minibatches
            for _e in range(num_updates_per_batch):
                perm = torch.randperm(n, device=dev, generator=gen)
                for k in range(num_minibatches):
                    idx = perm[k * mb:(k + 1) * mb]
                    params = nets.policy_network(obs_f[idx])
                    logp = dist.log_prob(params, raw_f[idx])
                    ratio = torch.exp(logp - logp_f[idx])
                    a = adv_f[idx]
                    policy_loss = -torch.minimum(ratio * a, ratio.clamp(1 - clipping_epsilon, 1 + clipping_epsilon) * a).mean()
                    v = nets.value_network(obs_f[idx]).squeeze(-1)
                    value_loss = 0.5 * 0.5 * ((tgt_f[idx] - v) ** 2).mean()
                    loss = policy_loss + value_loss
                    if entropy_cost:
                        loss = loss - entropy_cost * dist.entropy(params, gen).mean()
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    opt.step()
        run_eval(steps_done, t0)
    return result
