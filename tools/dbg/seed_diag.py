"""Where does a fuzz seed's one-step error come from?  Per env: forward outputs of the GPU vs the fp64 / fp32 oracle.
python tools/dbg/seed_diag.py <seed>"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_models import random_inputs, random_model
from test_gpu_kat import Gpu
from oracle.oracle import Oracle
seed = int(sys.argv[1])
dev = torch.device("cuda", 0)
mj = random_model(seed)
x0, us = random_inputs(mj, seed, 8, 5)
x0, us = x0.astype(np.float32), us.astype(np.float32)
print("nq nv nu nbody", mj.nq, mj.nv, mj.nu, mj.nbody, "neq", mj.neq, "npair", mj.npair, "pair_fn", list(mj.pair_fn), "jnt_type", list(mj.jnt_type), "limited", list(mj.jnt_limited),
      "solver", mj.opt.solver, "iters", mj.opt.iterations, mj.opt.ls_iterations, "cone/impratio", mj.opt.impratio)
g = Gpu(mj, dev)
o64, o32 = Oracle(mj, np.float64), Oracle(mj, np.float32)
ref = o64.rollout(x0, us); r32 = o32.rollout(x0, us); xs = g.rollout(x0, us)
for e in range(8):
    e1 = np.abs(xs[e, 1] - ref[e, 1]).max(); f1 = np.abs(r32[e, 1] - ref[e, 1]).max()
    f = g.forward(x0[e, :mj.nq], x0[e, mj.nq:], us[e, 0])
    o64.set_state(x0[e, :mj.nq], x0[e, mj.nq:], us[e, 0]); o64.forward()
    o32.set_state(x0[e, :mj.nq], x0[e, mj.nq:], us[e, 0]); o32.forward()
    line = [f"env {e}: step err gpu {e1:.2e} cpu32 {f1:.2e} nefc gpu {f['nefc']} oracle {int(o64.get('nefc')[0])} iters {int(o64.get('solver_niter')[0])}"]
    for k, name in (("qacc_smooth", "qacc_smooth"), ("qfrc_bias", "qfrc_bias"), ("qfrc_constraint", "qfrc_constraint"), ("qacc", "qacc")):
        a = o64.get(name); b = o32.get(name)
        line.append(f"{k} gpu {np.abs(f[k] - a).max():.1e}/cpu32 {np.abs(b - a).max():.1e} (scale {np.abs(a).max():.1e})")
    print("  ".join(line))
e = int(np.argmax([np.abs(xs[k, 1] - ref[k, 1]).max() for k in range(8)]))
d1, d32 = xs[e, 1] - ref[e, 1], r32[e, 1] - ref[e, 1]
print("worst env", e, "component errors gpu (qpos | qvel):")
print(np.array2string(d1[:mj.nq], precision=2), "|", np.array2string(d1[mj.nq:], precision=2))
print("cpu32:")
print(np.array2string(d32[:mj.nq], precision=2), "|", np.array2string(d32[mj.nq:], precision=2))
print("ref qvel after step", np.array2string(ref[e, 1, mj.nq:], precision=3))
