#!/usr/bin/env python3
"""Soak run on the GPU box: the headline rollout launched N times back to back; every launch's trajectories must be bit-identical to
the first's (same inputs), finite, and the launch time steady.  python tools/soak.py [launches] [model.xml]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import HEADLINE, make_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
HEADLINE = sys.argv[2] if len(sys.argv) > 2 else HEADLINE
mj = load_mj_model_from_file(HEADLINE)
model = mjx.put_model(mj, dev)
x0, us = make_inputs(mj, 4096, 1000, 1, 2, dev)
first = mjx.rollout(model, x0, us)
torch.cuda.synchronize()
assert bool(torch.isfinite(first).all())
times, bad = [], 0
for k in range(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xs = mjx.rollout(model, x0, us); b.record(); torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
    bad += int(not torch.equal(xs, first))
    if (k + 1) % 50 == 0:
        print(f"{k + 1} launches: {bad} differing from the first, last 50 launches {min(times[-50:]):.2f} .. {max(times[-50:]):.2f} ms", flush=True)
t = torch.tensor(times)
print(f"soak: {n} launches of {os.path.basename(HEADLINE)} 4096 x 1000, {bad} not bit-identical to the first, ms min / median / max = {t.min():.2f} / {t.median():.2f} / {t.max():.2f}")
sys.exit(1 if bad else 0)
