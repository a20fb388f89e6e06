"""A batch with an odd number of environments: the last wave's second lane group has no environment (its LDS slab is never
initialised).  The live environments must come out exactly as in an even batch (the fused Newton solve reads its wave mate's
slab for the matrix cores' other block since r03)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
for path in ("models/biped/biped.xml", "models/quadruped/quadruped.xml"):
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, 38, 50, 1, 2, dev)
    ref = mjx.rollout(model, x0, us)
    for n in (1, 5, 17, 33, 37):
        xs = mjx.rollout(model, x0[:n].contiguous(), us[:n].contiguous())
        same = torch.equal(xs, ref[:n])
        print(path.split("/")[-1], n, "bit-identical to the even batch:", same, "finite:", bool(torch.isfinite(xs).all()))
        assert same
