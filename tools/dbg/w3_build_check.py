import sys, os; sys.path.insert(0,'/root/repo')
import torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs, HEADLINE
dev=torch.device("cuda",0)
mj=load_mj_model_from_file(HEADLINE); model=mjx.put_model(mj, dev)
x0,us=make_inputs(mj,4096,1000,1,2,dev)
for w3 in ("0","1"):
    os.environ["AMX_FORCE_W3"]=w3
    mjx.rollout(model,x0,us); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): mjx.rollout(model,x0,us)
    b.record(); torch.cuda.synchronize(); ms=a.elapsed_time(b)/3
    print(f"AMX_FORCE_W3={w3}: {ms:.2f} ms {4096*1000/ms*1e-3:.2f} M/s")
