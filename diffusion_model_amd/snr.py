"""Learned noise schedule gamma(t) with the reference's parameter names (SNR.py:5-64).  It is
tabulated over the T+1 grid once per parameter version (diffusion.E3DiffusionProcess), so it never
sits on the per-step path; its three tiny dense layers use torch ops."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class PositiveLinear(nn.Module):
    """Linear layer with softplus-positive weights, no bias (SNR.py:5-22)."""

    def __init__(self, input_size, output_size, param_init_offset=-2.0):
        super().__init__()
        w = torch.empty(output_size, input_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w + param_init_offset)
        self.register_parameter("bias", None)
        self.param_init_offset = param_init_offset

    def forward(self, x):
        # SNR.py:21 wraps softplus(weight) in a fresh nn.Parameter, which detaches it from autograd
        # (only gamma_0 / gamma_1 ever train); reproduced with .detach().
        return F.linear(x, F.softplus(self.weight).detach(), self.bias)


class GammaNetwork(nn.Module):
    """Monotone gamma(t) of the VDM construction (SNR.py:27-64)."""

    def __init__(self):
        super().__init__()
        self.l1 = PositiveLinear(1, 1)
        self.l2 = PositiveLinear(1, 1024)
        self.l3 = PositiveLinear(1024, 1)
        self.gamma_0 = nn.Parameter(torch.tensor([-5.0]))
        self.gamma_1 = nn.Parameter(torch.tensor([10.0]))

    def gamma_tilde(self, t):
        l1_t = self.l1(t)
        return l1_t + self.l3(torch.sigmoid(self.l2(l1_t)))

    def forward(self, t):
        g0 = self.gamma_tilde(torch.zeros_like(t))
        g1 = self.gamma_tilde(torch.ones_like(t))
        gt = self.gamma_tilde(t)
        return self.gamma_0 + (self.gamma_1 - self.gamma_0) * ((gt - g0) / (g1 - g0))
