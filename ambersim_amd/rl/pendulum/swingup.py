"""Pendulum swing-up env (counterpart of reference ambersim/rl/pendulum/swingup.py)."""
import math
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Optional, Union

import torch

from ambersim_amd import mjx
from ambersim_amd.rl.base import MjxEnv, State
from ambersim_amd.utils.io_utils import load_mj_model_from_file


@dataclass
class PendulumSwingupConfig:
    """Same fields and defaults as the reference config (swingup.py:14-36)."""
    model_path: Union[Path, str] = "models/pendulum/scene.xml"
    physics_steps_per_control_step: int = 1
    stdev_obs: float = 0.0
    theta_cost_weight: float = 1.0
    theta_dot_cost_weight: float = 0.1
    control_cost_weight: float = 0.001
    qpos_hi: float = math.pi
    qpos_lo: float = -math.pi
    qvel_hi: float = 2
    qvel_lo: float = -2


class PendulumSwingupEnv(MjxEnv):
    """x = (theta, dtheta); obs = (cos theta, sin theta, dtheta); action = motor torque (reference :39-49)."""

    def __init__(self, config: Optional[PendulumSwingupConfig] = None, num_envs: int = 1, device=None) -> None:
        self.config = config if config is not None else PendulumSwingupConfig()
        mj_model = load_mj_model_from_file(self.config.model_path)
        super().__init__(mj_model, self.config.physics_steps_per_control_step, num_envs, device)

    @property
    def graph_capturable(self) -> bool:
        """step() issues device work only (no draw from a host-side generator): it can be captured in a HIP graph."""
        return self.config.stdev_obs == 0.0

    def fused_tail(self, episode_length: int):
        """Observation and reward of this env as the descriptor of a fused env.step (mjx.env_step -> amx_env_step: the physics launch
        and ONE elementwise launch for observation, reward, episode bookkeeping and auto reset), or None when a step draws
        observation noise from a host-side generator.  The same formulas as compute_obs / compute_reward below."""
        if self.config.stdev_obs != 0.0:
            return None
        # the fused step bypasses step / compute_obs / compute_reward entirely and knows no termination of its own: a subclass that
        # overrides any of them must not inherit this descriptor silently (it overrides fused_tail too, or runs the torch path)
        cls = type(self)
        if any(getattr(cls, n) is not getattr(PendulumSwingupEnv, n) for n in ("step", "compute_obs", "compute_reward")):
            return None
        c = self.config
        return mjx.make_env_tail([("cos", 0), ("sin", 0), ("qvel", 0)],
                                 [("wrapped_angle", 0, math.pi, c.theta_cost_weight), ("qvel", 0, 0.0, c.theta_dot_cost_weight),
                                  ("ctrl", 0, 0.0, c.control_cost_weight)], episode_length)

    def compute_obs(self, data: mjx.Data, info: Dict[str, Any]) -> torch.Tensor:
        theta = data.qpos[..., 0]
        return torch.stack((torch.cos(theta), torch.sin(theta), data.qvel[..., 0]), dim=-1)

    def compute_reward(self, data: mjx.Data, info: Dict[str, Any]) -> torch.Tensor:
        """-w_theta*wrap(theta-pi)^2 - w_dtheta*dtheta^2 - w_u*tau^2, maximal at theta=pi (reference :68-87)."""
        theta, theta_dot, tau = data.qpos[..., 0], data.qvel[..., 0], data.ctrl[..., 0]
        err = theta - math.pi
        err = torch.atan2(torch.sin(err), torch.cos(err))
        c = self.config
        return -c.theta_cost_weight * err ** 2 - c.theta_dot_cost_weight * theta_dot ** 2 - c.control_cost_weight * tau ** 2

    def reset(self, rng: torch.Generator) -> State:
        """qpos ~ U(qpos_lo, qpos_hi), qvel ~ U(qvel_lo, qvel_hi) per env (reference :89-104)."""
        c, dev, n = self.config, self.sys.device, self.num_envs
        u = lambda lo, hi, k: lo + (hi - lo) * torch.rand((n, k), generator=rng, device=dev, dtype=torch.float32)
        data = self.pipeline_init(u(c.qpos_lo, c.qpos_hi, self.sys.nq), u(c.qvel_lo, c.qvel_hi, self.sys.nv))
        obs = self.compute_obs(data, {})
        reward, done = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        return State(data, obs, reward, done, {"reward": reward}, {"rng": rng, "step": torch.zeros((), device=dev)})

    def step(self, state: State, action: torch.Tensor) -> State:
        """physics + observation (+ noise) + reward; done is always 0 (reference :106-122)."""
        rng = state.info["rng"]
        data = self.pipeline_step(state.pipeline_state, action)
        obs = self.compute_obs(data, state.info)
        if self.config.stdev_obs != 0.0:
            obs = obs + torch.randn(obs.shape, generator=rng, device=obs.device, dtype=obs.dtype) * self.config.stdev_obs
        reward = self.compute_reward(data, state.info)
        # functional update: the caller's State (and the dicts it holds) is left as it was - a captured graph re-reads its inputs
        info = dict(state.info, step=state.info["step"] + 1)  # a device counter: no host arithmetic inside a step
        return state.replace(pipeline_state=data, obs=obs, reward=reward, done=torch.zeros_like(reward), metrics=dict(state.metrics, reward=reward), info=info)
