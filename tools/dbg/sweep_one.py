"""One draw of tools/fuzz_sweep.py in detail (generic kernels): per-environment errors at steps 1 .. 5 against the fp64 oracle and the fp32 oracle's own.
  [AMX_LIBRARY=other.so] python tools/dbg/sweep_one.py <seed> <convex 0|1>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("AMX_SPECIALISE", "0")
import numpy as np, torch
from ambersim_amd import mjx
from fuzz_models import random_inputs, random_model
from oracle.oracle import Oracle
seed, convex = int(sys.argv[1]), bool(int(sys.argv[2]))
mj = random_model(seed, convex)
x0, us = random_inputs(mj, seed, 8, 5)
x0, us = x0.astype(np.float32), us.astype(np.float32)
ref = Oracle(mj, np.float64).rollout(x0, us)
ref32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
model = mjx.put_model(mj, torch.device("cuda", 0))
xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
print("library", os.environ.get("AMX_LIBRARY", "default"), "lanes", model._h.query("lanes_per_env"), "specialised", model._h.query("specialised"), "nv", mj.nv, "ncon", mj.ncon, "pairs", mj.pair_fn.tolist(),
      "solver", mj.opt.solver, "it", mj.opt.iterations, "ls", mj.opt.ls_iterations)
for t in range(1, 6):
    e = np.abs(xs[:, t] - ref[:, t]).max(axis=1) / max(1.0, np.abs(ref[:, t]).max())
    f = np.abs(ref32[:, t] - ref[:, t]).max(axis=1) / max(1.0, np.abs(ref[:, t]).max())
    print(f"step {t}: gpu " + " ".join(f"{v:.1e}" for v in e) + "   | fp32 oracle " + " ".join(f"{v:.1e}" for v in f))
