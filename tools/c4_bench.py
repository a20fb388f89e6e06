"""C4 (BASELINE.json configs[3]) alone: 1024 rollouts x 64 knots of the biped, forward-difference gradient of the terminal cost
through amx_fd_grad, repeated; used under rocprofv3 for profiles/rNN_rocprofv3_kernel_stats_c4_fd_grad.csv.
python tools/c4_bench.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ambersim_amd import mjx
from ambersim_amd.trajopt.fd import fd_eps, fd_grad_terminal_cost
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mj = load_mj_model_from_file("models/biped/biped.xml")
m = mjx.put_model(mj, dev)
x0, _ = make_inputs(mj, 1, 1, 1, 2, dev)
S, N, nx = 1024, 64, mj.nq + mj.nv
lo = torch.tensor(mj.actuator_ctrlrange[:, 0], dtype=torch.float32, device=dev); hi = torch.tensor(mj.actuator_ctrlrange[:, 1], dtype=torch.float32, device=dev)
us = 0.5 * (lo + hi) + 0.25 * (hi - lo) * (2 * torch.rand((N, mj.nu), generator=torch.Generator(device=dev).manual_seed(4), device=dev) - 1)
Qf = torch.eye(nx, device=dev)
call = lambda: fd_grad_terminal_cost(m, x0[0], us, Qf, x0[0], fd_eps(m), nsamples=S)
call(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): r = call()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
print(f"C4 biped: {S} rollouts x {N} steps, {S - 1} coordinates per launch: {ms:.3f} ms per gradient launch, {S * N / ms * 1e-3:.2f} M env-steps/s, grad finite {bool(torch.isfinite(r.grad).all())}")
