"""Quadratic trajectory cost (counterpart of reference ambersim/trajopt/cost.py:13-85; grad/hess are out of scope,
SURVEY.md §2 row 3)."""
import ctypes as C
from typing import Tuple

import torch

from ambersim_amd import _lib
from ambersim_amd.trajopt.base import CostFunction, CostFunctionParams


class StaticGoalQuadraticCost(CostFunction):
    """0.5*[sum_{t<N}(x_t-xg)'Q(x_t-xg) + (x_N-xg)'Qf(x_N-xg) + sum_t u_t'R u_t]."""

    def __init__(self, Q: torch.Tensor, Qf: torch.Tensor, R: torch.Tensor, xg: torch.Tensor) -> None:
        self.Q, self.Qf, self.R, self.xg = Q, Qf, R, xg

    def cost(self, xs: torch.Tensor, us: torch.Tensor, params: CostFunctionParams = None) -> Tuple[torch.Tensor, CostFunctionParams]:
        """xs [N+1,nx] (or [S,N+1,nx]), us [N,nu] (or [S,N,nu]) -> scalar (or [S]).  One HIP launch for a batch."""
        batched = xs.dim() == 3
        x = (xs if batched else xs.unsqueeze(0)).to(torch.float32).contiguous()
        u = (us if batched else us.unsqueeze(0)).to(torch.float32).contiguous()
        if not x.is_cuda:
            raise RuntimeError("StaticGoalQuadraticCost.cost runs on the GPU; move xs/us to the device")
        if x.dim() != 3 or u.dim() != 3:
            raise ValueError(f"cost: xs must be [N+1,nx] or [S,N+1,nx] and us [N,nu] or [S,N,nu], got {tuple(xs.shape)} and {tuple(us.shape)}")
        S, N, nx, nu = x.shape[0], u.shape[1], x.shape[2], u.shape[2]
        if u.device != x.device:
            raise ValueError(f"cost: xs on {x.device}, us on {u.device}")
        if u.shape[0] != S or x.shape[1] != N + 1:
            raise ValueError(f"cost: xs {tuple(xs.shape)} and us {tuple(us.shape)} do not describe the same rollouts (need [S,N+1,nx] and [S,N,nu])")
        f = lambda t: torch.as_tensor(t).to(device=x.device, dtype=torch.float32).contiguous()
        Q, Qf, R, xg = f(self.Q), f(self.Qf), f(self.R), f(self.xg)
        if Q.shape != (nx, nx) or Qf.shape != (nx, nx) or R.shape != (nu, nu) or xg.shape != (nx,):
            raise ValueError(f"cost: Q {tuple(Q.shape)}, Qf {tuple(Qf.shape)}, R {tuple(R.shape)}, xg {tuple(xg.shape)} do not match nx={nx}, nu={nu}")
        out = torch.empty(S, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(_lib.lib().amx_quadcost(S, N, nx, nu, x.data_ptr(), u.data_ptr(), Q.data_ptr(), Qf.data_ptr(), R.data_ptr(),
                                               xg.data_ptr(), out.data_ptr(), stream))
        return (out if batched else out[0]), params
