"""Abstract trajectory-optimisation API (counterpart of reference ambersim/trajopt/base.py; interface only)."""
from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass
class TrajectoryOptimizerParams:
    """Empty parameter base (reference trajopt/base.py:12-36)."""


class TrajectoryOptimizer:
    def optimize(self, params: TrajectoryOptimizerParams) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns (xs_star, us_star) (reference trajopt/base.py:64-78)."""
        raise NotImplementedError


@dataclass
class CostFunctionParams:
    """Empty parameter base (reference trajopt/base.py:86-88)."""


class CostFunction:
    def cost(self, xs: torch.Tensor, us: torch.Tensor, params: CostFunctionParams) -> Tuple[torch.Tensor, CostFunctionParams]:
        """Cost of one trajectory or of a batch (reference trajopt/base.py:105-120)."""
        raise NotImplementedError
