"""Env physics shim (counterpart of reference ambersim/rl/base.py).

The reference's env is written for one environment and batched by brax's vmap wrapper; torch has no
vmap over a ctypes call, so the env here is natively batched: every tensor carries a leading
[num_envs] axis, which is exactly the layout the vmapped reference produces."""
from abc import ABC, abstractmethod
from typing import Any, Dict, Optional

import torch

from ambersim_amd import mjx
from ambersim_amd.mjcf import MjData, MjModel


class State:
    """Environment state (reference ambersim/rl/base.py:14-32): pipeline_state, obs, reward, done, metrics, info."""

    def __init__(self, pipeline_state, obs, reward, done, metrics: Optional[Dict[str, Any]] = None, info: Optional[Dict[str, Any]] = None):
        self.pipeline_state = pipeline_state
        self.obs = obs
        self.reward = reward
        self.done = done
        self.metrics = {} if metrics is None else metrics
        self.info = {} if info is None else info

    def replace(self, **kw) -> "State":
        out = State(self.pipeline_state, self.obs, self.reward, self.done, self.metrics, self.info)
        for k, v in kw.items():
            if not hasattr(out, k):
                raise AttributeError(k)
            setattr(out, k, v)
        return out


class MjxEnv(ABC):
    """API for a batched physics env (reference ambersim/rl/base.py:35-153)."""

    def __init__(self, mj_model: MjModel, physics_steps_per_control_step: int = 1, num_envs: int = 1, device=None) -> None:
        assert physics_steps_per_control_step >= 1
        self.model = mj_model
        self.data = MjData(mj_model)
        self.sys = mjx.put_model(mj_model, device)
        self.num_envs = int(num_envs)
        self._physics_steps_per_control_step = physics_steps_per_control_step

    @property
    def dt(self) -> float:
        """Timestep of one env step (reference :56-58)."""
        return self.sys.opt.timestep * self._physics_steps_per_control_step

    @property
    def observation_size(self) -> int:
        """Dimension of the observation space, probed through reset (reference :61-65)."""
        return self.reset(torch.Generator(device=self.sys.device).manual_seed(0)).obs.shape[-1]

    @property
    def action_size(self) -> int:
        return self.sys.nu

    @property
    def backend(self) -> str:
        return "mjx"

    def pipeline_init(self, qpos: torch.Tensor, qvel: torch.Tensor) -> mjx.Data:
        """device data <- (qpos, qvel, ctrl=0), then one forward (reference :81-86)."""
        # the device image of the host MjData is made once per env (eleven small host-to-device copies otherwise, on every reset);
        # Data is functional - replace() returns a new record - so sharing the template's untouched leaves is safe
        if getattr(self, "_data_template", None) is None:
            self._data_template = mjx.device_put(self.data, self.sys.device)
            self._zero_ctrl = torch.zeros(self.sys.nu, dtype=torch.float32, device=self.sys.device)
        data = self._data_template.replace(qpos=qpos, qvel=qvel, ctrl=self._zero_ctrl)  # unbatched leaves broadcast to qpos' batch
        data = mjx.forward(self.sys, data)
        return data

    def pipeline_step(self, data: mjx.Data, ctrl: torch.Tensor) -> mjx.Data:
        """`physics_steps_per_control_step` physics steps with ctrl held (reference :88-96): one fused launch."""
        return mjx.step(self.sys, data.replace(ctrl=ctrl), self._physics_steps_per_control_step)

    def compute_reward(self, data: mjx.Data, info: Dict[str, Any]) -> torch.Tensor:
        raise NotImplementedError

    def compute_obs(self, data: mjx.Data, info: Dict[str, Any]) -> torch.Tensor:
        raise NotImplementedError

    @abstractmethod
    def reset(self, rng: torch.Generator) -> State:
        """Resets all envs."""

    @abstractmethod
    def step(self, state: State, action: torch.Tensor) -> State:
        """Steps all envs."""
