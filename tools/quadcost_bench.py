"""Diagnostic: amx_quadcost (batched StaticGoalQuadraticCost) throughput against its HBM roofline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ambersim_amd.trajopt.cost import StaticGoalQuadraticCost
dev = torch.device("cuda", 0)
for S, N, nx, nu in ((4096, 1000, 37, 12), (1024, 64, 37, 12), (4096, 1000, 61, 24), (100, 10, 16, 4)):
    g = torch.Generator(device=dev).manual_seed(0)
    xs = torch.randn((S, N + 1, nx), device=dev, generator=g); us = torch.randn((S, N, nu), device=dev, generator=g)
    A = torch.randn((nx, nx), device=dev, generator=g); Q = A @ A.T; B = torch.randn((nu, nu), device=dev, generator=g); R = B @ B.T
    cost = StaticGoalQuadraticCost(Q=Q, Qf=2 * Q, R=R, xg=torch.randn(nx, device=dev, generator=g))
    c = cost.batch_cost(xs, us) if hasattr(cost, "batch_cost") else None
    f = (lambda: cost.batch_cost(xs, us)) if c is not None else (lambda: cost.cost(xs, us))
    out = f(); torch.cuda.synchronize()
    # reference in torch (fp64)
    xc = (xs - cost.xg).double(); ref = 0.5 * (torch.einsum("sti,ij,stj->s", xc[:, :-1], Q.double(), xc[:, :-1]) + torch.einsum("si,ij,sj->s", xc[:, -1], (2 * Q).double(), xc[:, -1])
                                                + torch.einsum("sti,ij,stj->s", us.double(), R.double(), us.double()))
    out0 = out[0] if isinstance(out, tuple) else out
    err = ((out0.double() - ref).abs() / ref.abs().clamp(min=1)).max().item()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): f()
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    nbytes = 4 * (xs.numel() + us.numel())
    print(f"S={S} N={N} nx={nx} nu={nu}: {ms:.3f} ms  {nbytes / ms * 1e-6:.1f} GB/s of 8000 ({nbytes / ms * 1e-6 / 8000 * 100:.2f} %)  max rel err vs fp64 einsum {err:.2e}")
