#!/usr/bin/env python3
"""Wide fuzz sweep on the GPU box (not part of the test suite: minutes of CPU oracle time): both model families of tests/fuzz_models.py
over a range of seeds, same bars as tests/test_gpu_parity.py::test_random_models_vs_oracle / test_random_convex_models_vs_oracle.
  python tools/fuzz_sweep.py [first_seed] [last_seed]   -> one line per failing draw, a summary line per family"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("AMX_SPECIALISE", "0")
import numpy as np, torch
from ambersim_amd import mjx
from fuzz_models import random_inputs, random_model
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 260)
for convex in (False, True):
    tested = bad = 0
    worst = 0.0
    for seed in range(lo, hi):
        try:
            mj = random_model(seed, convex)
        except Exception as e:
            print(f"convex={convex} seed {seed}: model compiler raised {type(e).__name__}: {str(e)[:120]}", flush=True)
            continue
        if mj is None or (convex and mj.ncon == 0):
            continue
        x0, us = random_inputs(mj, seed, 8, 5)
        x0, us = x0.astype(np.float32), us.astype(np.float32)
        ref = Oracle(mj, np.float64).rollout(x0, us)
        ref32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
        if not (np.isfinite(ref).all() and np.isfinite(ref32).all()) or np.abs(ref).max() > 100:
            continue
        prng = np.random.default_rng(1000 + seed)
        sens = np.zeros(len(x0))
        for _ in range(3):
            xp = (x0 * (1 + 3e-7 * prng.standard_normal(x0.shape))).astype(np.float32)
            rp = Oracle(mj, np.float64).rollout(xp, us)
            sens = np.maximum(sens, np.abs(np.nan_to_num(rp, nan=1e30) - ref).reshape(len(ref), -1).max(axis=1) / max(1.0, np.abs(ref).max()))
        keep = sens < 1e-3
        if keep.sum() < 4:
            continue
        x0, us, ref, ref32 = x0[keep], us[keep], ref[keep], ref32[keep]
        try:
            model = mjx.put_model(mj, dev)
            xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
        except Exception as e:
            print(f"convex={convex} seed {seed}: {type(e).__name__}: {str(e)[:160]}", flush=True)
            bad += 1
            continue
        tested += 1
        def per_env(a, b, t):
            return np.abs(a[:, t] - b[:, t]).max(axis=1) / np.maximum(1.0, np.abs(b[:, t]).max())
        for t, tol, mult in ((1, 1e-5, 3), (5, 5e-5, 5)):
            e64, e32, f = per_env(xs, ref, t), per_env(xs, ref32, t), per_env(ref32, ref, t)
            ok = (e64 <= np.maximum(tol, mult * f)) | (e32 <= tol)
            if t == 1:
                worst = max(worst, float(np.minimum(e64, e32).max()))
            if not (np.isfinite(xs).all() and ok.all()):
                bad += 1
                print(f"convex={convex} seed {seed} step {t}: e64 {e64.max():.2e} e32 {e32.max():.2e} fp32-oracle floor {f.max():.2e} pairs {mj.pair_fn.tolist()} nv {mj.nv}", flush=True)
                break
    print(f"family convex={convex}: seeds {lo}..{hi - 1}, {tested} models run, {bad} outside the bars, worst one-step error {worst:.2e}", flush=True)
