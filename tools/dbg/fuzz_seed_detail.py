import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ.setdefault("AMX_SPECIALISE","0")
import numpy as np, torch
from ambersim_amd import mjx
from fuzz_models import *
from oracle.oracle import Oracle
dev=torch.device("cuda",0)
np.set_printoptions(linewidth=200, precision=3)
convex = "--plain" not in sys.argv
for seed in [int(s) for s in sys.argv[1:] if not s.startswith("--")]:
    mj = random_model(seed, convex)
    x0, us = random_inputs(mj, seed, 8, 5); x0=x0.astype(np.float32); us=us.astype(np.float32)
    ref = Oracle(mj, np.float64).rollout(x0, us); ref32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
    ref32k = Oracle(mj, np.float32, ls_rule=3).rollout(x0, us).astype(np.float64)
    model = mjx.put_model(mj, dev)
    xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
    sc = max(1.0, np.abs(ref[:,1]).max())
    pe = lambda a,b: np.abs(a[:,1]-b[:,1]).max(axis=1)/sc
    print("seed", seed, "pairs", mj.pair_fn.tolist(), "scale", sc)
    print("  gpu-fp64 ", pe(xs,ref)); print("  gpu-fp32 ", pe(xs,ref32)); print("  gpu-fp32k", pe(xs,ref32k)); print("  fp32-fp64", pe(ref32,ref)); print("  fp32k-fp64", pe(ref32k,ref))
    k = int(np.argmax(pe(xs,ref)))
    d = np.abs(xs[k,1]-ref[k,1]); j=int(np.argmax(d)); print("  worst env", k, "index", j, "(nq", mj.nq, ") gpu", xs[k,1,j], "fp64", ref[k,1,j], "fp32", ref32[k,1,j])
