import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from fuzz_models import random_inputs, random_model
from ambersim_amd import mjx
for seed in (26,):
    mj = random_model(seed)
    x0, us = random_inputs(mj, seed, 8, 5)
    x0, us = x0.astype(np.float32), us.astype(np.float32)
    for g in ("", "32", "64"):
        if g: os.environ["AMX_FORCE_G"] = g
        else: os.environ.pop("AMX_FORCE_G", None)
        model = mjx.put_model(mj, torch.device("cuda", 0))
        xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy()
        bad = ~np.isfinite(xs)
        print("seed", seed, "G", model._h.query("lanes_per_env"), "nv", mj.nv, "nbody", mj.nbody, "neq", mj.neq, "solver", mj.opt.solver, "iters", mj.opt.iterations, mj.opt.ls_iterations,
              "ncon", model.ncon, "nefc", model.nefc_max, "nonfinite envs", np.unique(np.where(bad)[0]), "first bad step", (np.where(bad)[1].min() if bad.any() else None))
