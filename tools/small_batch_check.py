import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
for path in ("models/quadruped/quadruped.xml",):
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev)
    for nenv, N in ((1024, 64), (2048, 64), (4096, 200)):
        x0, us = make_inputs(mj, nenv, N, 1, 2, dev)
        mjx.rollout(model, x0, us); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5): mjx.rollout(model, x0, us)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        print(f"G={model._h.query('lanes_per_env')} {nenv}x{N}: {ms:.3f} ms  {nenv*N/ms*1e-3:.2f} M env-steps/s")
