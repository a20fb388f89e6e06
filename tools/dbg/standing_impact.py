"""One-step parity of the resting biped (shipped kernels, one contact-row slot) started standing with its soles a little in the floor:
how far the GPU and the fp32 oracle are from the fp64 oracle per lowering depth (context for the 24-slot test's scenario)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import BIPED_REST, BIPED_STAND, make_inputs
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
from bench import solver_variant
CASES = ((BIPED_STAND, None), (BIPED_REST, None)) if os.environ.get("AMX_LIBRARY") else ((BIPED_REST, None), (BIPED_STAND, None), (BIPED_STAND, (10, 10)), (BIPED_STAND, (30, 30)))
for path, its in CASES:
    mj = load_mj_model_from_file(path)
    if its:
        mj = solver_variant(mj, *its)
    print("solver iterations", mj.opt.iterations, mj.opt.ls_iterations)
    model = mjx.put_model(mj, dev)
    for low in (0.0, 0.019, 0.021, 0.025, 0.03):
        xb, ub = make_inputs(mj, int(os.environ.get("NENV", "32")), 3, 1, 2, torch.device("cpu"))
        xb[:, 2] -= low
        x0, us = xb.numpy(), ub.numpy()
        xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
        ref = Oracle(mj, np.float64).rollout(x0, us); ref32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64); ref32k = Oracle(mj, np.float32, ls_rule=3).rollout(x0, us).astype(np.float64)
        pe = lambda a, b: (np.abs(a[:, 1] - b[:, 1]).max(axis=1) / np.maximum(1.0, np.abs(b[:, 1]).max(axis=1)))
        e = lambda a, b: pe(a, b).max()
        if os.environ.get("NENV"):
            q = lambda a, b: " / ".join(f"{np.quantile(pe(a, b), x):.2e}" for x in (0.5, 0.9, 1.0))
            print(f"{os.path.basename(path)} lowered {low}: p50 / p90 / max  gpu-fp64 {q(xs, ref)}   fp32-fp64 {q(ref32, ref)}   over 1e-5: gpu {int((pe(xs, ref) > 1e-5).sum())} fp32 {int((pe(ref32, ref) > 1e-5).sum())} of {len(x0)}", flush=True)
            continue
        print(f"{os.path.basename(path)} lowered {low}: gpu-fp64 {e(xs, ref):.2e}  gpu-fp32 {e(xs, ref32):.2e}  gpu-fp32k {e(xs, ref32k):.2e}  fp32-fp64 {e(ref32, ref):.2e}  fp32k-fp64 {e(ref32k, ref):.2e}", flush=True)
