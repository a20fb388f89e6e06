"""Episode bookkeeping for natively batched envs: what brax's EpisodeWrapper + AutoResetWrapper do around the
reference's env inside `ppo.train` (reference examples/rl/pendulum/ex_swingup.py:25-43 passes `episode_length`)."""
from typing import Optional

import torch

from ambersim_amd.rl.base import MjxEnv, State


def _blend(mask: torch.Tensor, new, old):
    if new is None or old is None or not torch.is_tensor(new):
        return old
    m = mask.view((-1,) + (1,) * (new.dim() - 1)).to(torch.bool)
    return torch.where(m, new, old)


class EpisodeAutoReset:
    """Truncates episodes at `episode_length` control steps and resets finished envs in place.

    `info["truncation"]` marks envs cut by the time limit (their value target bootstraps), `info["steps"]` counts
    control steps, `done` is the env's own termination OR the truncation.  A finished env restarts from the state it
    was first reset to, as brax's AutoResetWrapper does (`first_pipeline_state`, `first_obs`)."""

    def __init__(self, env: MjxEnv, episode_length: int, action_repeat: int = 1, fused: Optional[bool] = None):
        """fused (default: whenever the env offers a `fused_tail` descriptor, lives on a GPU and action_repeat is 1): env.step and
        this wrapper's bookkeeping run as one call of the C ABI (amx_env_step: the physics launch + one elementwise launch) instead
        of the env's and the wrapper's torch code (dozens of small kernels per step; the reference's env.step is one XLA program).
        AMX_FUSED_ENV_STEP=0 turns it off (A/B, and the reference point of tests/test_gpu_parity.py::test_fused_env_step...)."""
        import os
        self.env, self.episode_length, self.action_repeat = env, int(episode_length), int(action_repeat)
        tail = None
        if fused is not False and self.action_repeat == 1 and os.environ.get("AMX_FUSED_ENV_STEP", "1") != "0" and hasattr(env, "fused_tail") \
                and torch.device(env.sys.device).type == "cuda":
            tail = env.fused_tail(self.episode_length)
        if fused and tail is None:
            raise ValueError("this env offers no fused step (no fused_tail descriptor, observation noise, action_repeat > 1 or no GPU)")
        self._tail = tail

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, rng: torch.Generator) -> State:
        s = self.env.reset(rng)
        n, dev = s.reward.shape[0], s.reward.device
        s.info["steps"] = torch.zeros(n, device=dev)
        s.info["truncation"] = torch.zeros(n, device=dev)
        s.info["first_pipeline_state"] = s.pipeline_state
        s.info["first_obs"] = s.obs
        s.info["episode_return"] = torch.zeros(n, device=dev)
        s.info["last_episode_return"] = torch.zeros(n, device=dev)
        s.info["episodes"] = torch.zeros(n, device=dev)
        s.info["terminal_obs"] = s.obs.clone()  # present from the start: the State keeps one structure (graph capture)
        return s

    def step(self, state: State, action: torch.Tensor) -> State:
        """Functional: `state` and its dicts are not modified (the PPO loop replays this code as a HIP graph on persistent
        buffers; an in-place edit of the input's info would alias the graph's inputs and outputs)."""
        if self._tail is not None:
            return self._step_fused(state, action)
        info = dict(state.info)
        total = torch.zeros_like(state.reward)
        s = state
        for _ in range(self.action_repeat):
            s = self.env.step(s, action)
            total = total + s.reward
        info.update({k: v for k, v in s.info.items() if k not in info or k in ("step", "rng")})
        steps = state.info["steps"] + self.action_repeat
        trunc = (steps >= self.episode_length).to(s.reward.dtype)
        done = torch.maximum(s.done, trunc)
        fin = done > 0
        ret = state.info["episode_return"] + total
        info["last_episode_return"] = torch.where(fin, ret, state.info["last_episode_return"])
        info["episodes"] = state.info["episodes"] + done
        info["episode_return"] = torch.where(fin, torch.zeros_like(ret), ret)
        info["truncation"] = trunc * (1 - s.done)
        info["steps"] = torch.where(fin, torch.zeros_like(steps), steps)
        # auto reset: finished envs continue from their first state; the transition keeps this step's reward and done
        first, cur = state.info["first_pipeline_state"], s.pipeline_state
        # only the fields a step reads: the derived ones (qacc_smooth, qfrc_bias, xpos, ...) are not recomputed by mjx.step and are
        # stale pass-throughs in `cur` either way
        fields = {f: _blend(done, getattr(first, f), getattr(cur, f)) for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}
        data = cur.replace(**{k: v for k, v in fields.items() if v is not None})
        info["terminal_obs"] = s.obs  # observation before the reset, for bootstrapping truncated episodes
        obs = _blend(done, state.info["first_obs"], s.obs)
        return s.replace(pipeline_state=data, obs=obs, reward=total, done=done, info=info)

    def _step_fused(self, state: State, action: torch.Tensor) -> State:
        """The same transition through mjx.env_step; functional like step() (fresh output tensors, `state` untouched)."""
        from ambersim_amd import mjx
        env, si = self.env, state.info
        counters = {k: si[k] for k in ("steps", "episode_return", "last_episode_return", "episodes")}
        if torch.is_tensor(si.get("step")):
            counters["step"] = si["step"]
        data, o = mjx.env_step(env.sys, state.pipeline_state, action, env._physics_steps_per_control_step, self._tail,
                               si["first_pipeline_state"], si["first_obs"], counters)
        info = dict(si, steps=o["steps"], truncation=o["truncation"], episode_return=o["episode_return"], last_episode_return=o["last_episode_return"],
                    episodes=o["episodes"], terminal_obs=o["terminal_obs"])
        if "step" in o:
            info["step"] = o["step"]
        return state.replace(pipeline_state=data, obs=o["obs"], reward=o["reward"], done=o["done"], metrics=dict(state.metrics, reward=o["reward"]), info=info)
