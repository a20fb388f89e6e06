"""Diagnostic: parity vs the CPU oracle + rollout timing for the library selected by AMX_LIBRARY."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
print("library:", os.environ.get("AMX_LIBRARY", "default"))
for path, nenv, N in (("models/pendulum/scene.xml", 64, 100), ("models/quadruped/quadruped.xml", 64, 20), ("models/biped/biped.xml", 64, 20)):
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, nenv, N, 1, 2, dev)
    if mj.nq == 1:
        x0[:, 0] = torch.linspace(-3, 3, nenv); us = us * 0
    xs = mjx.rollout(model, x0, us).cpu().numpy()
    ref = Oracle(mj, np.float32).rollout(x0.cpu().numpy(), us.cpu().numpy(), nthreads=8)
    for k in (1, 5, N):
        print(f"  {path.split('/')[-1]:14s} step {k:3d} max|gpu-oracle32| {np.abs(xs[:, k] - ref[:, k]).max():.3e} (scale {np.abs(ref[:, k]).max():.1f})")
for path, nenv, N in (("models/quadruped/quadruped.xml", 4096, 200), ("models/biped/biped.xml", 4096, 100)):
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, nenv, N, 1, 2, dev)
    mjx.rollout(model, x0, us); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); xs = mjx.rollout(model, x0, us); t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    q = model._h.query
    print(f"  {path.split('/')[-1]:14s} {nenv}x{N}: {ms:.2f} ms -> {nenv*N/ms*1e3/1e6:.2f} M env-steps/s  finite={bool(torch.isfinite(xs).all())} "
          f"lds={q('lds_bytes')} slab={q('env_slab_bytes')} const={q('const_bytes')} envs/block={q('envs_per_block')} G={q('lanes_per_env')}")
