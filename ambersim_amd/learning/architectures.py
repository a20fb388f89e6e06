"""Network architectures (counterpart of reference ambersim/learning/architectures.py:7-34)."""
from typing import Sequence

import torch
from torch import nn


class MLP(nn.Module):
    """A simple pickle-able multi-layer perceptron.

    Args mirror the reference: `layer_sizes` are the sizes of all hidden layers and the output layer,
    `activate_final` applies the activation to the output too, `bias` toggles the linear biases.  torch modules need
    their input width up front, so it is inferred lazily on the first call (flax infers it the same way).
    """

    def __init__(self, layer_sizes: Sequence[int], activate_final: bool = False, bias: bool = True):
        super().__init__()
        self.layer_sizes = tuple(int(s) for s in layer_sizes)
        self.activate_final = activate_final
        self.bias = bias
        self.layers = nn.ModuleList([nn.LazyLinear(s, bias=bias) for s in self.layer_sizes])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i != len(self.layers) - 1 or self.activate_final:
                x = torch.relu(x)
        return x
