import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
Q = "models/quadruped/quadruped.xml"
for path, nenv, hor in [(Q, n, 200) for n in (4096, 5120, 6144, 7168, 8192, 9216, 10240, 12288)]:
    mj = load_mj_model_from_file(path); model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, nenv, hor, 1, 2, dev)
    mjx.rollout(model, x0, us); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): mjx.rollout(model, x0, us)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(os.path.basename(os.environ.get("AMX_LIBRARY", "")), path.split("/")[-1], nenv, hor, "%.2f ms %.1f M/s" % (ms, nenv * hor / ms * 1e-3))
