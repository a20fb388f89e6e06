"""`mjx`-shaped surface of the engine: Model, Data, device_put, make_data, forward, step.

Counterpart of the `mujoco.mjx` names the reference calls (reference ambersim/utils/io_utils.py:225-226,
ambersim/rl/base.py:52,83,85,93, ambersim/trajopt/shooting.py:34,36,41).  `Model` is shared and
unbatched; every `Data` leaf is a float32 torch tensor that may carry one leading batch axis
[nenv, ...] (the layout `jax.vmap` gives the reference).  The arithmetic runs in the HIP library
behind include/amx.h; there is no eager/CPU path here.
"""
from ._src import (Data, DisableBit, Model, Option, SolverType, constraint_activity, device_put, env_step, forward, make_data, make_env_tail, put_data,
                   put_model, rollout, step)

__all__ = ["Data", "DisableBit", "Model", "Option", "SolverType", "constraint_activity", "device_put", "env_step", "make_env_tail", "forward", "make_data", "put_data",
           "put_model", "rollout", "step"]
