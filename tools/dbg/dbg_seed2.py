import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from fuzz_models import random_inputs, random_model
from ambersim_amd import mjx
from oracle.oracle import Oracle
seed = 26
mj = random_model(seed)
x0, us = random_inputs(mj, seed, 8, 5)
x0, us = x0.astype(np.float32), us.astype(np.float32)
model = mjx.put_model(mj, torch.device("cuda", 0))
rng = np.random.default_rng(0)
for trial in range(6):
    xp = x0.copy()
    if trial: xp = (xp * (1 + 3e-7 * rng.standard_normal(xp.shape))).astype(np.float32)
    xs = mjx.rollout(model, torch.tensor(xp), torch.tensor(us)).cpu().numpy()
    r64 = Oracle(mj, np.float64).rollout(xp, us); r32 = Oracle(mj, np.float32).rollout(xp, us)
    print("trial", trial, "gpu max|x| per env", np.abs(xs).max(axis=(1, 2)).round(2)[4], "finite", np.isfinite(xs[4]).all(), "| f64 env4 max", np.abs(r64[4]).max().round(2), "f32", np.abs(r32[4]).max().round(2),
          "| env4 per step gpu", np.abs(xs[4]).max(axis=1).round(1), "f64", np.abs(r64[4]).max(axis=1).round(1))
