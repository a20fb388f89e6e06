"""env-steps/s of the falls-and-rests biped's fused rollout (bench.py's "C3 falls-and-rests" secondary alone), for A/B runs of the planner's
dense form: AMX_NO_DENSE=1 python tools/dbg/rest_rate.py  vs  python tools/dbg/rest_rate.py   [nenv horizon reps]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from ambersim_amd import mjx
from ambersim_amd import _lib

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hor = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
name = sys.argv[4] if len(sys.argv) > 4 else "models/biped/biped_rest.xml"
mj = load_mj_model_from_file(name)
t0 = time.time()
m = mjx.put_model(mj)
h = getattr(m, "handle", None) or getattr(m, "_h", None)
try:
    q = {k: (h or m).query(k) for k in ("lanes_per_env", "envs_per_block", "env_slab_bytes", "specialised")}
except Exception as e:
    q = repr(e)
print("setup %.1f s" % (time.time() - t0), "plan:", q)
g = torch.Generator(device="cuda").manual_seed(0)
lo, hi = torch.tensor(mj.actuator_ctrlrange[:, 0], device="cuda", dtype=torch.float32), torch.tensor(mj.actuator_ctrlrange[:, 1], device="cuda", dtype=torch.float32)
us = lo + (hi - lo) * torch.rand(nenv, hor, mj.nu, device="cuda", generator=g)
x0 = torch.tensor(np.concatenate([mj.qpos0, np.zeros(mj.nv)]), device="cuda", dtype=torch.float32)
xs = mjx.rollout(m, x0, us)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t = time.time(); xs = mjx.rollout(m, x0, us); torch.cuda.synchronize(); ts.append(time.time() - t)
print("finite", bool(torch.isfinite(xs).all()), "base z final mean %.4f" % float(xs[:, -1, 2].mean()))
print("ms %.2f  env-steps/s %.2f M" % (1e3 * min(ts), nenv * hor / min(ts) / 1e6))
