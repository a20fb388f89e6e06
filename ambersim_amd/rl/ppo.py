"""A torch PPO trainer that drives `MjxEnv.step`.

The reference trains with brax's `ppo.train` (reference examples/rl/pendulum/ex_swingup.py:25-43); brax and jax are
absent here, so this is a counterpart with the same knobs and the same batching arithmetic: every training step
collects `batch_size * num_minibatches * unroll_length` transitions from `num_envs` parallel envs, computes GAE
(lambda 0.95) on unrolls of `unroll_length`, and runs `num_updates_per_batch` epochs of `num_minibatches` clipped-
surrogate updates (clip 0.3, value coefficient 0.5) with Adam.  Observations are normalised by running statistics,
rewards scaled by `reward_scaling`, actions are tanh-squashed Gaussians.  Everything stays on the env's device; the
only physics call is `env.step` (one fused `amx_step` launch per control step for all envs)."""
import math
import os
import time
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

from ambersim_amd.rl.helpers import PPONetworks, make_ppo_networks
from ambersim_amd.rl.wrappers import EpisodeAutoReset


class RunningStatistics:
    """Running mean / std of observations (brax `running_statistics`), clipped like brax's normaliser.  All tensors are
    updated IN PLACE (a captured HIP graph holds their addresses): `mean_f` / `std_f` are the float32 copies `normalize` reads."""

    def __init__(self, size: int, device=None, max_abs_value: float = 5.0, min_std: float = 1e-6):
        self.count = torch.zeros((), device=device, dtype=torch.float64)
        self.mean = torch.zeros(size, device=device, dtype=torch.float64)
        self.m2 = torch.zeros(size, device=device, dtype=torch.float64)
        self.mean_f = torch.zeros(size, device=device, dtype=torch.float32)
        self.std_f = torch.full((size,), float(min_std), device=device, dtype=torch.float32)
        self.max_abs_value, self.min_std = max_abs_value, min_std

    def _refresh(self):
        self.mean_f.copy_(self.mean)
        self.std_f.copy_(self.std)

    def update(self, batch: torch.Tensor):
        b = batch.reshape(-1, batch.shape[-1]).to(torch.float64)
        n = b.shape[0]
        tot = self.count + n
        delta = b.mean(0) - self.mean
        self.m2.add_(((b - b.mean(0)) ** 2).sum(0) + delta ** 2 * self.count * n / tot)
        self.mean.add_(delta * n / tot)
        self.count.copy_(tot)
        self._refresh()

    @property
    def std(self):
        return torch.sqrt(self.m2 / torch.clamp(self.count, min=1.0)).clamp(min=self.min_std)

    def normalize(self, obs: torch.Tensor) -> torch.Tensor:
        out = (obs - self.mean_f.to(obs.dtype)) / self.std_f.to(obs.dtype)
        return out.clamp(-self.max_abs_value, self.max_abs_value)

    def state_dict(self):
        return {"count": self.count, "mean": self.mean, "m2": self.m2}

    def load_state_dict(self, d):
        self.count.copy_(d["count"]); self.mean.copy_(d["mean"]); self.m2.copy_(d["m2"])
        self._refresh()


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


# --------------------------------------------------------------------------------------------- device-resident stepping
# The reference's training step is one fused device program: brax jit-compiles the unroll (policy + env.step, scanned) and the
# SGD epochs, and nothing returns to the host in between (reference examples/rl/pendulum/ex_swingup.py:25-43 -> brax ppo.train;
# ambersim/rl/base.py:88-96 is a lax.scan).  The counterpart here: the same Python code is run ONCE under HIP-graph capture
# (torch.cuda.CUDAGraph) and replayed - `unroll_length` x {policy MLP, env.step (amx_step_to through the C ABI), observation,
# reward, episode bookkeeping, auto reset} is one graph launch, a minibatch update (forward, backward, Adam) another.  Eager
# execution remains (use_graphs=False, CPU tests of the host logic, envs that draw observation noise from a torch.Generator).
def _leaves(x, out=None):
    """Tensor leaves of a State / Data / dict / list, in a fixed order."""
    out = [] if out is None else out
    if torch.is_tensor(x):
        out.append(x)
    elif isinstance(x, dict):
        for k in sorted(x):
            _leaves(x[k], out)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _leaves(v, out)
    elif hasattr(x, "__slots__") and not isinstance(x, torch.Generator):
        for k in x.__slots__:
            _leaves(getattr(x, k), out)
    elif hasattr(x, "pipeline_state"):  # State
        for k in ("pipeline_state", "obs", "reward", "done", "metrics", "info"):
            _leaves(getattr(x, k), out)
    return out


def _persist(state):
    """A State whose tensor leaves are private contiguous buffers (the graphs read and rewrite them in place)."""
    def clone(x):
        if torch.is_tensor(x):
            return x.clone().contiguous()
        if isinstance(x, dict):
            return {k: clone(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(clone(v) for v in x)
        if hasattr(x, "__slots__") and not isinstance(x, torch.Generator):
            return type(x)(**{k: clone(getattr(x, k)) for k in x.__slots__})
        if hasattr(x, "pipeline_state"):
            return type(x)(clone(x.pipeline_state), clone(x.obs), clone(x.reward), clone(x.done), clone(x.metrics), clone(x.info))
        return x
    return clone(state)


def _write_back(dst, src):
    """dst leaves <- src leaves (same structure), in place; leaves that are already the same buffer are skipped."""
    a, b = _leaves(dst), _leaves(src)
    assert len(a) == len(b), "the env changed the structure of its State inside a step"
    for d, v in zip(a, b):
        if d.data_ptr() != v.data_ptr() or d.shape != v.shape:
            d.copy_(v.expand_as(d) if v.shape != d.shape else v)


class _Replayable:
    """`fn()` (no arguments, works on persistent buffers) run eagerly, or captured once as a HIP graph and replayed."""

    def __init__(self, fn: Callable[[], None], use_graph: bool, device, warmup: int = 0):
        self.fn, self.graph = fn, None
        if use_graph:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    fn()
            torch.cuda.current_stream(device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                fn()

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.fn()


def graphs_available(environment, device) -> bool:
    """HIP graphs apply when the env lives on a GPU and its step draws nothing from a host-side torch.Generator."""
    return torch.device(device).type == "cuda" and bool(getattr(environment, "graph_capturable", False))


class Evaluator:
    """`episode_length` steps of the deterministic (or sampled) policy in every env of `env`; the steps run in chunks of
    `chunk` captured as one graph."""

    def __init__(self, env, nets, norm, episode_length: int, deterministic: bool, use_graph: bool, chunk: int = 20):
        self.env, self.n = env, episode_length
        dev = env.sys.device
        dist = nets.parametric_action_distribution
        self.chunk = max(1, min(chunk, episode_length))
        while episode_length % self.chunk:
            self.chunk -= 1
        self.state = None
        self.total = None
        self.noise = None if deterministic else torch.zeros((self.chunk, env.num_envs, env.action_size), device=dev)

        def body():
            s = self.state
            tot = self.total
            for t in range(self.chunk):
                x = norm.normalize(s.obs) if norm is not None else s.obs
                params = nets.policy_network(x)
                if deterministic:
                    act = dist.mode(params)
                else:
                    loc, scale = dist._split(params)
                    act = dist.postprocess(loc + scale * self.noise[t])
                s = env.step(s, act)
                tot = tot + s.reward
            _write_back(self.state, s)
            self.total.copy_(tot)
        self._body, self._use_graph, self._run = body, use_graph, None

    def __call__(self, seed: int) -> float:
        dev = self.env.sys.device
        rng = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            s0 = self.env.reset(rng)
            if self.state is None:
                self.state = _persist(s0)
                self.total = torch.zeros_like(s0.reward)
                self._run = _Replayable(self._body, self._use_graph, dev, warmup=1 if self._use_graph else 0)
            _write_back(self.state, s0)
            self.total.zero_()
            for _ in range(self.n // self.chunk):
                if self.noise is not None:
                    self.noise.normal_(generator=rng)
                self._run()
        return float(self.total.mean().item())


def evaluate(env, policy, episode_length: int, seed: int) -> float:
    """Mean undiscounted return of `policy` over one episode in every env of `env` (brax's `eval/episode_reward`).  Eager."""
    rng = torch.Generator(device=env.sys.device).manual_seed(seed)
    s = env.reset(rng)
    total = torch.zeros_like(s.reward)
    for _ in range(episode_length):
        s = env.step(s, policy(s.obs))
        total = total + s.reward
    return float(total.mean().item())


class Collector:
    """One unroll: `unroll_length` x {policy, env.step} from every env, results in persistent [T, N, ...] buffers."""

    def __init__(self, env, nets, norm, unroll_length: int, reward_scaling: float, state, use_graph: bool):
        self.T = unroll_length
        dev = env.sys.device
        dist = nets.parametric_action_distribution
        N, A, O = state.obs.shape[0], env.action_size, state.obs.shape[-1]
        f = lambda *sh: torch.zeros(sh, device=dev)
        self.state = _persist(state)
        self.noise = f(self.T, N, A)
        self.obs, self.raw, self.logp, self.rew = f(self.T, N, O), f(self.T, N, A), f(self.T, N), f(self.T, N)
        self.done, self.trunc, self.nobs = f(self.T, N), f(self.T, N), f(self.T, N, O)

        def body():
            # per step only what the next step needs (normalise, policy, sample, env.step); the trajectory buffers are filled once
            # per unroll - one stack per field and ONE log-probability evaluation over [T, N] - instead of seven small copies and a
            # dozen elementwise kernels per step (r04: 96 -> 7x us per step of the captured unroll, profiles/r04_ppo_breakdown.txt)
            s = self.state
            obs_l, par_l, raw_l, rew_l, done_l, trunc_l, nobs_l = [], [], [], [], [], [], []
            for t in range(self.T):
                obs = s.obs
                x = norm.normalize(obs) if norm is not None else obs
                params = nets.policy_network(x)
                loc, scale = dist._split(params)
                raw = loc + scale * self.noise[t]
                obs_l.append(obs); par_l.append(params); raw_l.append(raw)
                s = env.step(s, dist.postprocess(raw))  # functional: fresh output tensors, the ones collected above stay as they are
                rew_l.append(s.reward); done_l.append(s.done); trunc_l.append(s.info["truncation"]); nobs_l.append(s.info["terminal_obs"])
            torch.stack(obs_l, out=self.obs); torch.stack(raw_l, out=self.raw)
            self.logp.copy_(dist.log_prob(torch.stack(par_l), self.raw))
            torch.stack(rew_l, out=self.rew); self.rew.mul_(reward_scaling)
            torch.stack(done_l, out=self.done); torch.stack(trunc_l, out=self.trunc); torch.stack(nobs_l, out=self.nobs)
            _write_back(self.state, s)
        self._run = _Replayable(body, use_graph, dev, warmup=0)  # no warm-up run: it would advance the envs (capture does not execute)

    def __call__(self, gen: torch.Generator):
        self.noise.normal_(generator=gen)
        self._run()
        return self


def train(environment, num_timesteps: int, episode_length: int, num_envs: int = 1, num_evals: int = 1, action_repeat: int = 1,
          unroll_length: int = 10, batch_size: int = 32, num_minibatches: int = 16, num_updates_per_batch: int = 2, learning_rate: float = 1e-4,
          entropy_cost: float = 1e-4, discounting: float = 0.9, reward_scaling: float = 1.0, normalize_observations: bool = False,
          clipping_epsilon: float = 0.3, gae_lambda: float = 0.95, seed: int = 0, network_factory: Callable = make_ppo_networks,
          progress_fn: Callable = lambda *a: None, eval_env=None, deterministic_eval: bool = True, use_graphs: Optional[bool] = None) -> PPOResult:
    """Argument names and defaults follow brax's `ppo.train` as called by the reference example.  `environment` must be
    built with `num_envs` parallel envs; `eval_env` (optional, same class) is used for the `num_evals` evaluations.
    `use_graphs` (default: whenever the env allows it, see graphs_available): unrolls, evaluation chunks and minibatch updates
    run as replayed HIP graphs - no host round trip inside an unroll, as in the reference's jitted training step."""
    env = EpisodeAutoReset(environment, episode_length, action_repeat)
    assert environment.num_envs == num_envs, "build the environment with num_envs parallel envs"
    assert (batch_size * num_minibatches) % num_envs == 0, "batch_size * num_minibatches must be a multiple of num_envs"
    dev = environment.sys.device
    if use_graphs is None:
        use_graphs = graphs_available(environment, dev)
    torch.manual_seed(seed)
    gen = torch.Generator(device=dev).manual_seed(seed)
    obs_size, act_size = environment.observation_size, environment.action_size
    nets = network_factory(obs_size, act_size)  # e.g. make_ppo_networks or PPONetworksWrapper(...).make_ppo_networks
    nets.policy_network.to(dev); nets.value_network.to(dev)
    dist = nets.parametric_action_distribution
    norm = RunningStatistics(obs_size, dev) if normalize_observations else None
    nets.policy_network(torch.zeros((1, obs_size), device=dev)); nets.value_network(torch.zeros((1, obs_size), device=dev))  # materialise lazy layers
    graph_updates = use_graphs and not entropy_cost  # the entropy estimate draws from the generator
    params_all = list(nets.policy_network.parameters()) + list(nets.value_network.parameters())
    # torch's single-kernel Adam (one launch instead of a dozen per update: 16.9 -> 15.2 s for the swing-up example, final rewards in
    # the same distribution over seeds - profiles/r03_ppo_seeds.txt); AMX_PPO_FUSED_ADAM=0 restores the per-tensor form
    fused = os.environ.get("AMX_PPO_FUSED_ADAM", "1") != "0" and dev.type == "cuda"
    opt = torch.optim.Adam(params_all, lr=learning_rate, capturable=bool(graph_updates), **({"fused": True} if fused else {}))

    env_steps_per_training_step = batch_size * unroll_length * num_minibatches * action_repeat
    num_evals_after_init = max(num_evals - 1, 1)
    training_steps_per_epoch = max(1, math.ceil(num_timesteps / (num_evals_after_init * env_steps_per_training_step)))
    unrolls_per_step = batch_size * num_minibatches // num_envs
    result = PPOResult(nets, norm, [])
    e_env = eval_env if eval_env is not None else environment
    evaluator = Evaluator(e_env, nets, norm, episode_length // action_repeat, deterministic_eval, use_graphs and graphs_available(e_env, dev))

    def run_eval(steps_done, t0):
        r = evaluator(seed + 1)
        m = {"eval/episode_reward": r, "walltime": time.time() - t0}
        result.metrics.append((steps_done, m))
        progress_fn(steps_done, m)

    t0 = time.time()
    with torch.no_grad():
        collector = Collector(env, nets, norm, unroll_length, reward_scaling, env.reset(gen), use_graphs)
    if num_evals > 1:
        run_eval(0, t0)

    # persistent batch buffers of one training step: [T, U*N, ...] -> flattened [n, ...]
    T, U, N = unroll_length, unrolls_per_step, num_envs
    n = T * U * N
    mb = n // num_minibatches
    f = lambda *sh: torch.zeros(sh, device=dev)
    obs_t, raw_t, logp_t, rew_t = f(T, U * N, obs_size), f(T, U * N, act_size), f(T, U * N), f(T, U * N)
    done_t, trunc_t, nobs_t = f(T, U * N), f(T, U * N), f(T, U * N, obs_size)
    obs_f, raw_f, logp_f, adv_f, tgt_f = f(n, obs_size), f(n, act_size), f(n), f(n), f(n)
    idx = torch.zeros(mb, dtype=torch.int64, device=dev)

    def update_body():
        """One clipped-surrogate minibatch update on the rows `idx` of the flattened batch."""
        o = obs_f.index_select(0, idx)
        params = nets.policy_network(o)
        logp = dist.log_prob(params, raw_f.index_select(0, idx))
        ratio = torch.exp(logp - logp_f.index_select(0, idx))
        a = adv_f.index_select(0, idx)
        policy_loss = -torch.minimum(ratio * a, ratio.clamp(1 - clipping_epsilon, 1 + clipping_epsilon) * a).mean()
        v = nets.value_network(o).squeeze(-1)
        value_loss = 0.5 * 0.5 * ((tgt_f.index_select(0, idx) - v) ** 2).mean()
        loss = policy_loss + value_loss
        if entropy_cost:
            loss = loss - entropy_cost * dist.entropy(params, gen).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    update, eager_updates = None, 0  # the graph is captured after three eager (real) updates on a side stream, as torch's capture recipe asks
    side = torch.cuda.Stream(device=dev) if graph_updates else None

    steps_done = 0
    for _ in range(num_evals_after_init):
        for _ in range(training_steps_per_epoch):
            # ---- collect unrolls_per_step x unroll_length transitions from every env
            with torch.no_grad():
                for u in range(U):
                    c = collector(gen)
                    sl = slice(u * N, (u + 1) * N)
                    obs_t[:, sl].copy_(c.obs); raw_t[:, sl].copy_(c.raw); logp_t[:, sl].copy_(c.logp); rew_t[:, sl].copy_(c.rew)
                    done_t[:, sl].copy_(c.done); trunc_t[:, sl].copy_(c.trunc); nobs_t[:, sl].copy_(c.nobs)
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
                obs_f.copy_(flat(nx(obs_t))); raw_f.copy_(flat(raw_t)); logp_f.copy_(flat(logp_t)); tgt_f.copy_(flat(targets))
                a_ = flat(adv)
                adv_f.copy_((a_ - a_.mean()) / (a_.std() + 1e-8))
            steps_done += env_steps_per_training_step
            # ---- updates: minibatches of batch_size * unroll_length transitions
            for _e in range(num_updates_per_batch):
                perm = torch.randperm(n, device=dev, generator=gen)
                for k in range(num_minibatches):
                    idx.copy_(perm[k * mb:(k + 1) * mb])
                    if not graph_updates:
                        update_body()
                    elif update is not None:
                        update()
                    elif eager_updates < 3:
                        side.wait_stream(torch.cuda.current_stream(dev))
                        with torch.cuda.stream(side):
                            update_body()
                        torch.cuda.current_stream(dev).wait_stream(side)
                        eager_updates += 1
                    else:
                        opt.zero_grad(set_to_none=True)
                        update = _Replayable(update_body, True, dev, warmup=0)  # capture executes nothing: replay it for this minibatch
                        update()
        run_eval(steps_done, t0)
    return result
