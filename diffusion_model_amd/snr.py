"""Learned noise schedule gamma(t) with the reference's parameter names (SNR.py:5-64).  It is
tabulated over the T+1 grid once per parameter version (diffusion.E3DiffusionProcess), so it never
sits on the per-step path.  On the GPU gamma_tilde runs in the library's kernel (egnn_gamma_tilde); the
softplus-positive weights carry no gradient in the reference (PositiveLinear wraps them in a fresh
Parameter), so only the final affine map to [gamma_0, gamma_1] is under autograd either way.  CPU tensors (host-side
tabulation, CPU tests) use torch ops."""
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

    def _gamma_tilde_device(self, t):
        """gamma_tilde of [t | 0 | 1] in one launch of the library's kernel (t [n, 1] on the GPU) -> (gt [n,1], g0, g1)"""
        from . import _lib
        n = t.shape[0]
        pts = torch.cat((t.detach().reshape(-1).float(), torch.tensor([0.0, 1.0], device=t.device))).contiguous()
        out = torch.empty(n + 2, device=t.device)
        w1, w2, w3 = (m.weight.detach().float().contiguous().reshape(-1) for m in (self.l1, self.l2, self.l3))
        _lib.check(_lib.lib().egnn_gamma_tilde(_lib.stream_ptr(), n + 2, w2.numel(), _lib.ptr(pts), _lib.ptr(w1), _lib.ptr(w2),
                                               _lib.ptr(w3), _lib.ptr(out)))
        return out[:n].view(n, 1), out[n], out[n + 1]

    def forward(self, t):
        # the library kernel evaluates gamma_tilde without a graph: only when no gradient with respect to t is asked for
        # (the reference detaches the softplus weights, SNR.py:21, but gamma(t) stays differentiable in t).  The kernel's
        # and the torch path's fp32 sums differ by up to ~5e-4 on gamma (as the reference itself does across thread counts,
        # DESIGN.md section 2): tabulate the training and the sampling schedule on the SAME device.
        if t.is_cuda and t.dim() == 2 and t.shape[1] == 1 and self.l2.weight.is_cuda and not (t.requires_grad and torch.is_grad_enabled()):
            gt, g0, g1 = self._gamma_tilde_device(t)
        else:
            g0 = self.gamma_tilde(torch.zeros_like(t))
            g1 = self.gamma_tilde(torch.ones_like(t))
            gt = self.gamma_tilde(t)
        return self.gamma_0 + (self.gamma_1 - self.gamma_0) * ((gt - g0) / (g1 - g0))
