#!/usr/bin/env python3
"""The three biped workloads side by side: parity against the fp64 CPU oracle on a small batch, throughput
at the bench shape, and what the robots physically do (pelvis height, contacts, constraint rows).

    python tools/variants_check.py [--nenv 4096] [--horizon 1000] [--out profiles/r04_biped_variants.txt]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ambersim_amd import mjx  # noqa: E402
from ambersim_amd.utils.io_utils import load_mj_model_from_file  # noqa: E402
from bench import activity_stats, make_inputs  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

VARIANTS = (("stands", "models/biped/biped_stand.xml"), ("falls-and-rests", "models/biped/biped_rest.xml"), ("legacy feet-only", "models/biped/biped.xml"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nenv", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lines = []
    for name, path in VARIANTS:
        mj = load_mj_model_from_file(path)
        model = mjx.put_model(mj, dev)
        q = model._h.query
        x0, us = make_inputs(mj, 64, 20, 1, 2, dev)
        xs = mjx.rollout(model, x0, us).cpu().numpy().astype(np.float64)
        ref = Oracle(mj, np.float64).rollout(x0.cpu().numpy(), us.cpu().numpy())
        err = lambda k: float((np.abs(xs[:, k] - ref[:, k]).max(axis=1) / np.maximum(1.0, np.abs(ref[:, k]).max(axis=1))).max())
        x0, us = make_inputs(mj, a.nenv, a.horizon, 1, 2, dev)
        mjx.rollout(model, x0, us)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(); xs_t = mjx.rollout(model, x0, us); t1.record(); torch.cuda.synchronize()
            ms.append(t0.elapsed_time(t1))
        st = activity_stats(model, mj, xs_t, us)
        lines.append(f"{name:<18}{os.path.basename(path):<18} G={q('lanes_per_env')} spec={q('specialised')} ncon={mj.ncon} | parity vs fp64 oracle, 64 envs: "
                     f"N=1 {err(1):.2e} N=5 {err(5):.2e} N=20 {err(20):.2e} | {a.nenv}x{a.horizon}: {np.median(ms):.2f} ms = "
                     f"{a.nenv * a.horizon / np.median(ms) / 1e3:.1f} M env-steps/s | " + " ".join(f"{k}={v:.3g}" if isinstance(v, float) else f"{k}={v}" for k, v in st.items()))
        print(lines[-1], flush=True)
        del xs_t, us
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
