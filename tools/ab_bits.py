#!/usr/bin/env python3
"""Are two development libraries (tools/devbuild.sh) bit-identical on the bench's inputs?  python tools/ab_bits.py libA.so libB.so [...]
Each library in its own process: sha256 of the rollout (AB_CASES models, 256 envs x 200 steps) - a change that only removes
instructions must print the same digests."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, hashlib, torch
sys.path.insert(0, %r)
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
out = []
for name in os.environ.get("AB_CASES", "biped_stand,biped").split(","):
    path = {"quadruped": "models/quadruped/quadruped.xml"}.get(name, "models/biped/%%s.xml" %% name)
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev, specialise=False)
    x0, us = make_inputs(mj, 256, 200, 1, 2, dev)
    xs = mjx.rollout(model, x0, us).cpu().numpy()
    out.append("%%s %%s finite=%%s" %% (name, hashlib.sha256(xs.tobytes()).hexdigest()[:16], bool(torch.isfinite(torch.from_numpy(xs)).all())))
print(" | ".join(out))
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, AMX_LIBRARY=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib):28s} {r.stdout.strip() or r.stderr[-400:]}", flush=True)
