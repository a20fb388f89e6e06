"""Finite-difference gradient of the terminal cost of a shooting rollout (BASELINE.json configs[3]).

The reference has no finite-difference code (its gradients differentiate the cost only, reference
ambersim/trajopt/base.py:138-139); this workload composes the reference's `shoot` (ambersim/trajopt/shooting.py:22-48)
and `StaticGoalQuadraticCost.cost` with Q = R = 0 (ambersim/trajopt/cost.py:62-85): 1 nominal + (S-1) perturbed rollouts
from one shared x0, each perturbing one control coordinate by +eps, forward differences.  One C-ABI call per launch
(`amx_fd_grad`: perturbation, fused rollout keeping only final states, cost + difference)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from ambersim_amd import _lib, mjx


@dataclass
class FDGradResult:
    J0: torch.Tensor      # () nominal terminal cost
    grad: torch.Tensor    # [ncoord] dJ/du over flat control coordinates coord0 .. coord0+ncoord-1 (t-major: c = t*nu + i)
    J: torch.Tensor       # [ncoord+1] costs of the nominal (index 0) and the perturbed rollouts
    eps: torch.Tensor     # [nu] perturbation per actuator
    xfinal: torch.Tensor  # [ncoord+1, nq+nv] final states


def fd_eps(m: mjx.Model, eps_rel: float = 1e-3) -> torch.Tensor:
    """eps_i = eps_rel * (ctrlrange width of actuator i) (eps = 1e-3 * range).  Validated on the HOST range
    (no device round trip) and cached per model and eps_rel."""
    cache = m.__dict__.setdefault("_fd_eps_cache", {})
    if eps_rel not in cache:
        import numpy as np
        lim = np.asarray(m.mj_model.actuator_ctrlrange, dtype=np.float64).reshape(-1, 2)
        if eps_rel <= 0 or bool((lim[:, 1] - lim[:, 0] <= 0).any()):
            raise ValueError("fd_eps: every actuator needs a ctrlrange of positive width (or pass eps explicitly)")
        eps = (eps_rel * (m.actuator_ctrlrange[:, 1] - m.actuator_ctrlrange[:, 0])).to(torch.float32).contiguous()
        eps._amx_positive = True  # checked on the host above
        cache[eps_rel] = eps
    return cache[eps_rel]


def _fd_buffers(m: mjx.Model, nsamples: int, N: int):
    """Workspace and outputs of one (nsamples, N) launch shape, allocated once per model: the C4 loop calls this path hundreds of
    times per second and an allocation + a workspace query per call sat inside its timed region."""
    cache = m.__dict__.setdefault("_fd_buf_cache", {})
    key = (nsamples, N)
    if key not in cache:
        L = m._h.L
        need = C.c_size_t()
        _lib.check(L.amx_fd_workspace_bytes(m._h.ptr, nsamples, N, C.byref(need)), L)
        cache[key] = torch.empty(max(need.value, 4) // 4 + 1, dtype=torch.float32, device=m.device)
        if len(cache) > 8:  # a handful of shapes at most: drop the oldest
            cache.pop(next(iter(cache)))
    return cache[key]


def fd_grad_terminal_cost(m: mjx.Model, x0: torch.Tensor, us_nominal: torch.Tensor, Qf: torch.Tensor, xg: torch.Tensor,
                          eps: Optional[torch.Tensor] = None, nsamples: int = 1024, coord0: int = 0) -> FDGradResult:
    """One launch: rollout 0 = nominal, rollout k perturbs flat control coordinate coord0+k-1 by +eps[i].

    x0 [nq+nv]; us_nominal [N,nu]; Qf [nx,nx]; xg [nx].  With N*nu > nsamples-1 only nsamples-1 coordinates are
    differenced per launch; `fd_grad_all` walks coord0 over the rest."""
    dev = m.device
    nx, nu = m.nq + m.nv, m.nu
    f = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.float32).contiguous()
    x0, us_nominal, Qf, xg = f(x0), f(us_nominal), f(Qf), f(xg)
    eps = fd_eps(m) if eps is None else f(eps)
    if x0.shape != (nx,) or us_nominal.dim() != 2 or us_nominal.shape[1] != nu or Qf.shape != (nx, nx) or xg.shape != (nx,) or eps.shape != (nu,):
        raise ValueError(f"fd_grad_terminal_cost: shapes x0 {tuple(x0.shape)}, us {tuple(us_nominal.shape)}, Qf {tuple(Qf.shape)}, "
                         f"xg {tuple(xg.shape)}, eps {tuple(eps.shape)} do not match nx={nx}, nu={nu}")
    if nsamples < 2 or coord0 < 0:
        raise ValueError("fd_grad_terminal_cost: need nsamples >= 2 and coord0 >= 0")
    if not getattr(eps, "_amx_positive", False):  # fd_eps() has validated its own result on the host; a caller's eps costs one sync, once
        if bool((eps <= 0).any()):
            raise ValueError("fd_grad_terminal_cost: eps must be positive (actuators need a ctrlrange, or pass eps)")
        eps._amx_positive = True
    N = us_nominal.shape[0]
    L = m._h.L  # the library build the model lives in
    ws = _fd_buffers(m, nsamples, N)  # reused across calls; stream-ordered, so back-to-back launches on one stream are safe
    J = torch.empty(nsamples, dtype=torch.float32, device=dev)
    g = torch.empty(nsamples - 1, dtype=torch.float32, device=dev)
    xf = torch.empty((nsamples, nx), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.amx_fd_grad(m._h.ptr, nsamples, N, x0.data_ptr(), us_nominal.data_ptr(), eps.data_ptr(), int(coord0), Qf.data_ptr(),
                                 xg.data_ptr(), J.data_ptr(), g.data_ptr(), xf.data_ptr(), ws.data_ptr(), ws.numel() * 4, stream))
    return FDGradResult(J0=J[0], grad=g, J=J, eps=eps, xfinal=xf)


def fd_grad_all(m: mjx.Model, x0, us_nominal, Qf, xg, eps=None, nsamples: int = 1024) -> torch.Tensor:
    """Gradient over ALL N*nu control coordinates, [N,nu], in ceil(N*nu / (nsamples-1)) launches."""
    us_nominal = torch.as_tensor(us_nominal)
    N, nu = us_nominal.shape
    out = []
    for c0 in range(0, N * nu, nsamples - 1):
        out.append(fd_grad_terminal_cost(m, x0, us_nominal, Qf, xg, eps, nsamples, c0).grad)
    return torch.cat(out)[: N * nu].reshape(N, nu)
