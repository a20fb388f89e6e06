#!/usr/bin/env python3
"""Arithmetic operations per env-step of the restated algorithm (the CPU restatement built with oracle/count_real.hpp, which counts
every +, -, *, /, sqrt, sin, cos on a REAL; dense algorithm) for the bench workloads -> profiles/rNN_flop_census.json, which bench.py
quotes as a RECORDED figure next to the measured throughput.  Runs on the CPU: python tools/flop_census.py [round]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
from oracle.oracle import Oracle
import torch

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = {}
for tag, path in (("quadruped", "models/quadruped/quadruped.xml"), ("biped", "models/biped/biped.xml"), ("biped_stand", "models/biped/biped_stand.xml"),
                  ("biped_rest", "models/biped/biped_rest.xml")):
    mj = load_mj_model_from_file(path)
    x0, us = make_inputs(mj, 4, 200, 1, 2, torch.device("cpu"))
    x0, us = x0.numpy(), us.numpy()
    oc = Oracle(mj, variant="count")
    oc.lib.amxo_flops_read_and_reset.restype = ctypes.c_double
    tot, n = 0.0, 0
    for e in range(4):
        oc.set_state(x0[e, :mj.nq], x0[e, mj.nq:], np.zeros(mj.nu), np.zeros(mj.nv))
        oc.forward()
        oc.lib.amxo_flops_read_and_reset()
        for t in range(200):
            oc.set_state(None, None, us[e, t], None)
            oc.step()
        tot += oc.lib.amxo_flops_read_and_reset(); n += 200
    out[tag] = tot / n
    print(tag, out[tag])
path = os.path.join(ROOT, "profiles", f"{rnd}_flop_census.json")
json.dump({"flops_per_env_step": out, "how": "tools/flop_census.py: counting build of the CPU restatement, first 4 envs x 200 steps of the bench workload, dense algorithm"},
          open(path, "w"), indent=1)
print(path)
