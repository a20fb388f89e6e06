"""Diagnostic: per-phase cycle shares of the rollout kernel (libamx_prof.so, built with -DAMX_PROFILE).
Run: AMX_LIBRARY=ambersim_amd/csrc/libamx_prof.so python tools/phase_profile.py [model] [nenv] [steps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ambersim_amd import _lib, mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs

NAMES = {0: "step head/ctrl", 1: "kinematics", 2: "com_pos+cinert+cdof", 3: "crb+M", 4: "collision", 5: "make_constraint",
         6: "com_vel+rne", 7: "smooth forces", 8: "factor M + qacc_smooth", 9: "solver init (warm/smooth cost)",
         10: "constraint force+grad", 11: "newton: chol H + solve", 12: "iter cond", 13: "linesearch", 14: "newton: assemble H",
         15: "solve tail", 16: "euler: damp solve", 17: "euler: integrate", 18: "write xs row", 19: "  H: copy M", 20: "  crb: subtree up-scan", 21: "  ls: Mv, Jv, gauss sums", 22: "  ls: bracketing loop", 23: "  vel: joint sums + down-scan", 24: "  newton: before the matrix cores", 25: "  newton: J'DJ on the matrix cores", 26: "  newton: row assembly", 27: "  newton: elimination",
         28: "  kin: joints to local poses", 29: "  kin: pose composition", 30: "  vel: joint sums", 31: "  com: tree COM"}

model_path = sys.argv[1] if len(sys.argv) > 1 else "models/quadruped/quadruped.xml"
nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda", 0)
mj = load_mj_model_from_file(model_path)
model = mjx.put_model(mj, dev)
x0, us = make_inputs(mj, nenv, steps, 1, 2, dev)
prof = torch.zeros((nenv, 32), dtype=torch.float32, device=dev)
L = _lib.lib()
L.amx_prof_set.argtypes = [C.c_void_p]
L.amx_prof_set(prof.data_ptr())
mjx.rollout(model, x0, us)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
prof.zero_()
t0.record(); mjx.rollout(model, x0, us); t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1)
p = prof.cpu().numpy().astype(np.float64)
tot = p.sum(axis=1).mean()
print(f"{model_path}: {nenv} envs x {steps} steps  kernel {ms:.2f} ms  -> {nenv*steps/ms*1e3/1e6:.2f} M env-steps/s")
print(f"kernel: {'specialised' if model._h.query('specialised') else 'generic'}, lane group {model._h.query('lanes_per_env')}, {model._h.query('envs_per_block')} envs per workgroup")
print(f"mean cycles per env-step per wave: {tot/steps:.0f}")
# load balance: a wave's time is fixed by its slower environment, the launch ends with the slowest wave of the slowest workgroup
per_env = p.sum(axis=1) / steps
G = model._h.query("lanes_per_env"); epw = 64 // G; epb = model._h.query("envs_per_block")
per_wave = per_env[: (nenv // epw) * epw].reshape(-1, epw).max(axis=1)
per_wg = per_env[: (nenv // epb) * epb].reshape(-1, epb).max(axis=1)
q = lambda a: " ".join(f"{np.percentile(a, x):.0f}" for x in (0, 10, 50, 90, 100))
print(f"cycles per step, percentiles 0/10/50/90/100 - per env: {q(per_env)} | per wave: {q(per_wave)} | per workgroup (max of its waves): {q(per_wg)}")
print(f"slowest workgroup / mean wave: {per_wg.max() / per_wave.mean():.3f}")
wpb = epb // epw
if wpb > 1 and nenv % epb == 0:  # by position of the wave inside its workgroup: waves w and w + 4 share a SIMD
    by_idx = per_wave.reshape(-1, wpb).mean(axis=0)
    print("mean cycles per step by wave index in the workgroup: " + " ".join(f"{v:.0f}" for v in by_idx))
for i in range(32):
    v = p[:, i].mean()
    if v > 0:
        print(f"  {i:2d} {NAMES.get(i, '?'):34s} {v/steps:10.0f} cyc/step  {100*v/tot:5.1f}%")
