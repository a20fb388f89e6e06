#!/usr/bin/env python3
"""Parity per horizon: for C1 (pendulum), C2 (quadruped) and C3 (biped), the
distance of the GPU fp32 rollout and of the CPU oracle's fp32 rollout to the CPU oracle's fp64 rollout, on the same
seeded inputs, at N = 1, 10, 100, 1000 steps.  Since r04 the fp32 CPU oracle appears twice: with MJX's line-search
formulation (ls_rule 0, the statement of record) and with the kernel's acceptance rule and arithmetic (ls_rule 3,
oracle/amx_oracle.c; tests/test_ls_rule.py) - the column the GPU is expected to sit on.

    python tools/parity_horizons.py [--nenv 256] [--out profiles/r02_parity_horizons.txt]

Per env e and horizon N: err_e = max_i |x_N[i] - ref_N[i]| / max(1, max_i |ref_N[i]|) over the state x = (qpos, qvel)
(north_star's "1e-5 relative on qpos/qvel after N steps").  Reported: the median and the maximum over envs.  The fp32
CPU column is the floor any fp32 implementation of the same algorithm sits on: contact dynamics amplify rounding
differences (the 4-point manifold selection and the activation of a constraint row are discontinuous in the state),
so beyond a few steps both fp32 columns leave 1e-5 together.  Needs a GPU; runs the oracle with all host threads."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HORIZONS = (1, 10, 100, 1000)
CONFIGS = (("C1 pendulum", "models/pendulum/pendulum.xml"), ("C2 quadruped", "models/quadruped/quadruped.xml"),
           ("C3 biped", "models/biped/biped.xml"), ("C3 stand", "models/biped/biped_stand.xml"), ("C3 rest", "models/biped/biped_rest.xml"))


def inputs(mj, nenv, N, seed):
    """BASELINE.md inputs: C1 qpos~U(-pi,pi), qvel~U(-2,2), zero ctrl; C2/C3 standing keyframe + N(0,0.01^2) velocity noise,
    ctrl ~ U(ctrlrange) i.i.d."""
    rng = np.random.default_rng(seed)
    if mj.nq == 1:
        x0 = np.stack([rng.uniform(-np.pi, np.pi, nenv), rng.uniform(-2, 2, nenv)], axis=1)
        return x0, np.zeros((nenv, N, mj.nu))
    q0 = mj.key_qpos[0] if mj.nkey else mj.qpos0
    x0 = np.tile(np.concatenate([q0, np.zeros(mj.nv)]), (nenv, 1))
    x0[:, mj.nq:] += 0.01 * rng.standard_normal((nenv, mj.nv))
    us = rng.uniform(mj.actuator_ctrlrange[:, 0], mj.actuator_ctrlrange[:, 1], (nenv, N, mj.nu))
    return x0.astype(np.float32).astype(np.float64), us.astype(np.float32).astype(np.float64)  # exactly representable in fp32


def env_err(xs, ref, N):
    a, b = xs[:, N].astype(np.float64), ref[:, N]
    return np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))


def measure(path, nenv, horizons=HORIZONS, seed=7, gpu=True):
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    from oracle.oracle import Oracle
    mj = load_mj_model_from_file(path)
    N = max(horizons)
    x0, us = inputs(mj, nenv, N, seed)
    ref = Oracle(mj, np.float64).rollout(x0, us)
    o32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
    k32 = Oracle(mj, np.float32, ls_rule=3).rollout(x0, us).astype(np.float64)
    rows = {}
    xg = None
    if gpu:
        import torch
        from ambersim_amd import mjx
        model = mjx.put_model(mj, torch.device("cuda", 0))
        xg = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy()
    for h in horizons:
        f = env_err(o32, ref, h)
        k = env_err(k32, ref, h)
        rows[h] = {"cpu_fp32_median": float(np.median(f)), "cpu_fp32_max": float(f.max()), "cpu_within_1e-5_frac": float((f <= 1e-5).mean()),
                   "cpu_krule_median": float(np.median(k)), "cpu_krule_max": float(k.max()), "cpu_krule_within_1e-5_frac": float((k <= 1e-5).mean())}
        if xg is not None:
            g = env_err(xg, ref, h)
            rows[h].update({"gpu_fp32_median": float(np.median(g)), "gpu_fp32_max": float(g.max()),
                            "gpu_within_1e-5_frac": float((g <= 1e-5).mean()),
                            # the GPU against the fp32 CPU oracle that runs the kernel's rule (not against fp64): how closely the two
                            # fp32 implementations of the SAME algorithm track each other
                            "gpu_vs_krule_median": float(np.median(env_err(xg, k32, h))), "gpu_vs_mjxrule_median": float(np.median(env_err(xg, o32, h)))})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nenv", type=int, default=256)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lines = [f"# parity per horizon: {args.nenv} envs per config, relative error of the state (qpos, qvel) vs the fp64 CPU oracle",
             "# err_e = max_i |x_N - ref_N| / max(1, max_i |ref_N|); median / max over envs; `<=1e-5` = fraction of envs within north_star's tolerance",
             "# CPU fp32 = the oracle in single precision with MJX's line-search rule; CPU fp32 k-rule = the same with the kernel's acceptance rule and arithmetic (ls_rule 3)",
             "# GPU-vs = median distance of the GPU rollout to either fp32 CPU rollout (not to fp64)",
             f"{'config':<14}{'N':>6} | {'GPU fp32 median':>16}{'GPU fp32 max':>14}{'<=1e-5':>8} | {'CPU fp32 median':>16}{'CPU fp32 max':>14}{'<=1e-5':>8} | "
             f"{'CPU k-rule median':>18}{'k-rule max':>14}{'<=1e-5':>8} | {'GPU-vs-k-rule':>14}{'GPU-vs-MJX-rule':>16}"]
    blob = {}
    for name, path in CONFIGS:
        rows = measure(path, args.nenv)
        blob[name] = rows
        for h, r in rows.items():
            lines.append(f"{name:<14}{h:>6} | {r['gpu_fp32_median']:>16.3e}{r['gpu_fp32_max']:>14.3e}{r['gpu_within_1e-5_frac']:>8.2f} | "
                         f"{r['cpu_fp32_median']:>16.3e}{r['cpu_fp32_max']:>14.3e}{r['cpu_within_1e-5_frac']:>8.2f} | "
                         f"{r['cpu_krule_median']:>18.3e}{r['cpu_krule_max']:>14.3e}{r['cpu_krule_within_1e-5_frac']:>8.2f} | "
                         f"{r['gpu_vs_krule_median']:>14.3e}{r['gpu_vs_mjxrule_median']:>16.3e}")
    text = "\n".join(lines)
    print(text)
    print(json.dumps(blob))
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
