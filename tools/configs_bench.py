"""Throughput of every BASELINE.md configuration and of the per-call step API on one GPU (diagnostic; bench.py is the
contract line).  Run from the repo root on a GPU box: python tools/configs_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs

dev = torch.device("cuda", 0)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def rollout_case(tag, path, nenv, horizon, reps, shared_x0=False):
    mj = load_mj_model_from_file(path)
    model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, nenv, horizon, 1, 2, dev)
    if shared_x0:
        x0 = x0[0].contiguous()
    ms = timed(lambda: mjx.rollout(model, x0, us), reps)
    nb = 4 * (mj.nu + mj.nq + mj.nv)
    print(f"{tag:44s} {nenv:6d} envs x {horizon:5d} steps  {ms:9.3f} ms/launch  {nenv * horizon / ms * 1e-3:8.2f} M env-steps/s  "
          f"{nenv * horizon * nb / ms * 1e-6:7.2f} GB/s algorithmic ({nb} B/env-step)  G={model._h.query('lanes_per_env')}")


def step_case(tag, path, nenv, nsub, reps):
    mj = load_mj_model_from_file(path)
    model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, nenv, 1, 1, 2, dev)
    data = mjx.make_data(model, nenv).replace(qpos=x0[:, : mj.nq].contiguous(), qvel=x0[:, mj.nq:].contiguous(), ctrl=us[:, 0].contiguous())
    state = {"d": mjx.forward(model, data)}

    def one():
        state["d"] = mjx.step(model, state["d"], nsub)
    ms = timed(one, reps)
    nb = 4 * (2 * (mj.nq + 2 * mj.nv) + mj.nu)
    print(f"{tag:44s} {nenv:6d} envs x {nsub:5d} substeps {ms:8.3f} ms/call    {nenv * nsub / ms * 1e-3:8.2f} M env-steps/s  "
          f"{nenv * nb / ms * 1e-6:7.2f} GB/s algorithmic ({nb} B/call/env)")


if __name__ == "__main__":
    Q, B, P = "models/quadruped/quadruped.xml", "models/biped/biped.xml", "models/pendulum/pendulum.xml"
    rollout_case("C1 pendulum rollout", P, 64, 100, 20)
    rollout_case("C2 quadruped rollout (bench.py workload)", Q, 4096, 1000, 3)
    rollout_case("C3 biped rollout", B, 4096, 1000, 2)
    rollout_case("C4 quadruped FD rollouts, shared x0", Q, 1024, 64, 20, shared_x0=True)
    rollout_case("C4 biped FD rollouts, shared x0", B, 1024, 64, 20, shared_x0=True)
    rollout_case("C5 shape per GPU: biped 4096 x 1000", B, 4096, 1000, 1)
    rollout_case("quadruped, 32768 envs x 200", Q, 32768, 200, 2)
    rollout_case("pendulum, 65536 envs x 1000", P, 65536, 1000, 2)
    step_case("A1 step API: quadruped, 1 substep", Q, 4096, 1, 50)
    step_case("A1 step API: biped, 1 substep", B, 4096, 1, 50)
    step_case("A1 step API: biped, 4 substeps", B, 4096, 4, 30)
    step_case("A1 step API: pendulum 1024 envs (PPO example)", P, 1024, 1, 100)
