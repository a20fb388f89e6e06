"""Seeded random CONTACT-RICH models (tests/fuzz_models.py: random_dense_model - 9 .. 16 sphere contacts, Newton, limits only) through the per-model
kernels of the planner's dense form against the fp64 oracle.  Step 1 here (no GPU): `python tools/dbg/fuzz_dense.py build <first> <count>` compiles
csrc/ for each model's plan (cached in csrc/jit/, which travels to the GPU box).  Step 2 on the GPU box: `... run <first> <count>`."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_models import random_dense_inputs, random_dense_model
mode, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rel = lambda a, b: np.abs(a - b).max(axis=-1) / np.maximum(1.0, np.abs(b).max(axis=-1))
if mode == "build":
    from ambersim_amd import _spec
    n = 0
    for seed in range(first, first + count):
        mj = random_dense_model(seed)
        if mj is None: continue
        p = _spec.plan(mj)
        _spec.build_specialised_library(mj); n += 1
        print(seed, "nv", p["nv"], "ncon", p["ncon"], "G", p["G"], "nbs", p["nbs"], "jfr", p["jfr"], "mcomp", p["mcomp"], "nb0", p["mnb0"], "stage", _spec._LAST_BUILD.get("stage"), flush=True)
    print("built", n)
else:
    import torch
    from ambersim_amd import mjx
    from oracle.oracle import Oracle
    dev = torch.device("cuda", 0)
    bad, worst, n, dense = [], 0.0, 0, 0
    for seed in range(first, first + count):
        mj = random_dense_model(seed)
        if mj is None: continue
        o = Oracle(mj, np.float64)
        x0, us = random_dense_inputs(mj, o, seed, 16, 5)
        x0, us = x0.astype(np.float32), us.astype(np.float32)
        ref = o.rollout(x0, us); ref32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
        if not (np.isfinite(ref).all() and np.isfinite(ref32).all()) or np.abs(ref).max() > 100: continue
        touching = 0
        for k in range(16):
            o.set_state(x0[k, :mj.nq], x0[k, mj.nq:], us[k, 0]); o.forward(); touching += int((o.get("con_dist") < 0).sum())
        model = mjx.put_model(mj, dev, specialise=True)
        assert model._h.query("specialised") == 1
        xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
        e1, f1, e5, f5 = rel(xs[:, 1], ref[:, 1]), rel(ref32[:, 1], ref[:, 1]), rel(xs[:, 5], ref[:, 5]), rel(ref32[:, 5], ref[:, 5])
        ok = bool(np.isfinite(xs).all() and (e1 <= np.maximum(1e-5, 3 * f1)).all() and (e5 <= np.maximum(5e-5, 5 * f5)).all())
        n += 1; dense += int(model._h.query("lanes_per_env") == 32 and model._h.query("envs_per_block") == 16)
        worst = max(worst, e1.max())
        print(f"seed {seed} nv {mj.nv} ncon {mj.ncon} lanes {model._h.query('lanes_per_env')} touching {touching}/16 envs: e1 {e1.max():.2e} (fp32 oracle {f1.max():.2e}) "
              f"e5 {e5.max():.2e} ({f5.max():.2e}) {'ok' if ok else 'FAIL'}", flush=True)
        if not ok: bad.append(seed)
    print(f"{n} models ({dense} on the dense form), worst one-step error {worst:.2e}, failures {bad}")
