"""Seeded random models through the PER-MODEL (specialised, static) kernels against the fp64 oracle - the generic kernels have
tools/dbg/fuzz_long.py.  Step 1 here (no GPU): `python tools/dbg/fuzz_spec.py build <first> <count>` compiles csrc/ for each model's plan
(about a minute each, cached in csrc/jit/, which travels to the GPU box).  Step 2 on the GPU box: `... run <first> <count>`."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_models import random_inputs, random_model
mode, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
if mode == "build":
    from ambersim_amd import _spec
    n = 0
    for seed in range(first, first + count):
        mj = random_model(seed)
        if mj is None: continue
        p = _spec.plan(mj)
        print(seed, "nv", p["nv"], "G", p["G"], "nefc", p["nefc"], "ncon", p["ncon"], "neq", p["neq"], "fusedH", p["fusedH"], "tree_H", p.get("tree_H"), flush=True)
        _spec.build_specialised_library(mj); n += 1
    print("built", n)
else:
    import torch
    from ambersim_amd import mjx
    from oracle.oracle import Oracle
    dev = torch.device("cuda", 0)
    bad = []
    for seed in range(first, first + count):
        mj = random_model(seed)
        if mj is None: continue
        x0, us = random_inputs(mj, seed, 8, 5)
        x0, us = x0.astype(np.float32), us.astype(np.float32)
        ref = Oracle(mj, np.float64).rollout(x0, us); ref32 = Oracle(mj, np.float32).rollout(x0, us)
        if not (np.isfinite(ref).all() and np.isfinite(ref32).all()) or np.abs(ref).max() > 100: continue
        model = mjx.put_model(mj, dev, specialise=True)
        gen = mjx.put_model(mj, dev)
        assert model._h.query("specialised") == 1
        xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
        xg = mjx.rollout(gen, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
        e1, f1, e5, f5 = rel(xs[:, 1], ref[:, 1]), rel(ref32[:, 1], ref[:, 1]), rel(xs[:, 5], ref[:, 5]), rel(ref32[:, 5], ref[:, 5])
        ok = np.isfinite(xs).all() and e1 <= max(1e-5, 3 * f1) and e5 <= max(5e-5, 5 * f5)
        print(f"seed {seed} nv {mj.nv} G {model._h.query('lanes_per_env')}: spec e1 {e1:.2e} (floor {f1:.2e}) e5 {e5:.2e} (floor {f5:.2e}); spec vs generic 5-step {rel(xs[:, 5], xg[:, 5]):.2e} {'ok' if ok else 'FAIL'}")
        if not ok: bad.append(seed)
    print("failures", bad)
