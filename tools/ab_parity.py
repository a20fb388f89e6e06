#!/usr/bin/env python3
"""Parity of a development library (AMX_LIBRARY, tools/devbuild.sh) against the fp64 CPU oracle on the models it holds:
python tools/ab_parity.py models/biped/biped_stand.xml [...]   - relative state error at N = 1, 5, 20 over 64 envs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
print("library:", os.environ.get("AMX_LIBRARY", "default"))
for path in sys.argv[1:]:
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev, specialise=False)
    x0, us = make_inputs(mj, 64, 20, 1, 2, dev)
    xs = mjx.rollout(model, x0, us).cpu().numpy().astype(np.float64)
    ref = Oracle(mj, np.float64).rollout(x0.cpu().numpy(), us.cpu().numpy())
    k32 = Oracle(mj, np.float32, ls_rule=3).rollout(x0.cpu().numpy(), us.cpu().numpy()).astype(np.float64)
    e = lambda a, b, k: float((np.abs(a[:, k] - b[:, k]).max(axis=1) / np.maximum(1.0, np.abs(b[:, k]).max(axis=1))).max())
    print(f"  {os.path.basename(path):18s} spec={model._h.query('specialised')} vs fp64: " + " ".join(f"N={k} {e(xs, ref, k):.2e}" for k in (1, 5, 20)) +
          " | cpu fp32 k-rule vs fp64: " + " ".join(f"N={k} {e(k32, ref, k):.2e}" for k in (1, 5, 20)))
