#!/usr/bin/env python3
"""A/B throughput of library variants: python tools/ab_bench.py libA.so libB.so ...  (each in its own process; C2 and C3)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
out = []
import os
CASES = (("models/quadruped/quadruped.xml", 4096, 1000), ("models/biped/biped_stand.xml", 4096, 1000), ("models/biped/biped.xml", 4096, 1000), ("models/biped/biped.xml", 1024, 64),
         ("models/quadruped/quadruped.xml", 32768, 200), ("models/biped/biped_stand.xml", 32768, 100), ("models/biped/biped_rest.xml", 4096, 1000))
if os.environ.get("AB_CASES"):  # e.g. AB_CASES=biped_stand,biped: only those models (a library built with SPECS=... holds no others)
    CASES = tuple(c for c in CASES if c[0].split("/")[-1][:-4] in os.environ["AB_CASES"].split(","))
for path, nenv, hor in CASES:
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, nenv, hor, 1, 2, dev)
    mjx.rollout(model, x0, us); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 if hor == 1000 else 20
    a.record()
    for _ in range(reps): mjx.rollout(model, x0, us)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    out.append("%%s %%dx%%d %%.2f ms %%.2f M/s" %% (path.split("/")[-1][:-4], nenv, hor, ms, nenv * hor / ms * 1e-3))
print(" | ".join(out))
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, AMX_LIBRARY=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib):28s} {r.stdout.strip() or r.stderr[-400:]}", flush=True)
