"""Optimizers of the reference's define_optimizer (parts/def_for_main.py:119-139): Adam, AdamW(amsgrad) from
torch, and RAdamScheduleFree.

RAdamScheduleFree comes from the third-party package ``schedulefree`` (un-pinned in the reference, absent
offline).  It is restated here from the published algorithm (Defazio et al., "The Road Less Scheduled",
2024, schedule-free wrapper of RAdam: interpolation y = (1 - beta1) z + beta1 x, Polyak-style averaging
weights c_{k+1} = lr_max^p / sum, RAdam rectification of the step size with a silent SGD phase).
PARITY UNPINNED: neither the package nor any fixture of it is available; the tests check the algebraic
properties only (train/eval round trip, equivalence with plain SGD-free averaging identities, convergence).
"""
from __future__ import annotations

import torch


class RAdamScheduleFree(torch.optim.Optimizer):
    def __init__(self, params, lr=0.0025, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, r=0.0,
                 weight_lr_power=2.0, silent_sgd_phase=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, r=r, weight_lr_power=weight_lr_power,
                        silent_sgd_phase=silent_sgd_phase, k=0, train_mode=False, weight_sum=0.0, lr_max=-1.0,
                        scheduled_lr=0.0)
        super().__init__(params, defaults)

    @torch.no_grad()
    def eval(self):
        """parameters hold y while training; switch them to the averaged iterate x (train_per_iretation.py:189-190)"""
        for group in self.param_groups:
            if group["train_mode"]:
                beta1, _ = group["betas"]
                for p in group["params"]:
                    st = self.state[p]
                    if "z" in st:
                        p.lerp_(end=st["z"].to(p.device), weight=1 - 1 / beta1)   # y -> x
                group["train_mode"] = False

    @torch.no_grad()
    def train(self):
        """x -> y (train_per_iretation.py:103-104)"""
        for group in self.param_groups:
            if not group["train_mode"]:
                beta1, _ = group["betas"]
                for p in group["params"]:
                    st = self.state[p]
                    if "z" in st:
                        p.lerp_(end=st["z"].to(p.device), weight=1 - beta1)       # x -> y
                group["train_mode"] = True

    @torch.no_grad()
    def step(self, closure=None):
        if not self.param_groups[0]["train_mode"]:
            raise RuntimeError("RAdamScheduleFree.step() called in eval mode: call optimizer.train() first")
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            eps, (beta1, beta2), decay = group["eps"], group["betas"], group["weight_decay"]
            step = group["k"] + 1
            beta2_t = beta2 ** step
            bias_correction2 = 1 - beta2_t
            rho_inf = 2 / (1 - beta2) - 1                         # maximum length of the approximated SMA
            rho_t = rho_inf - 2 * step * beta2_t / bias_correction2
            if rho_t > 4.0:
                rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
            else:
                rect = float(not group["silent_sgd_phase"])
            lr = group["scheduled_lr"] = group["lr"] * rect
            group["lr_max"] = lr_max = max(lr, group["lr_max"])
            weight = (step ** group["r"]) * (lr_max ** group["weight_lr_power"])
            weight_sum = group["weight_sum"] = group["weight_sum"] + weight
            ckp1 = weight / weight_sum if weight_sum != 0 else 0.0
            adaptive_y_lr = lr * (beta1 * (1 - ckp1) - 1)
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "z" not in st:
                    st["z"] = p.detach().clone(memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                z, v = st["z"], st["exp_avg_sq"]
                g = p.grad
                v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
                if rho_t > 4.0:
                    gn = g / v.div(bias_correction2).sqrt_().add_(eps)
                else:
                    gn = g.clone()
                if decay != 0:
                    gn.add_(p, alpha=decay)                         # decay at y
                p.lerp_(end=z, weight=ckp1)                         # y <- (1 - c) y + c z
                p.add_(gn, alpha=adaptive_y_lr)
                z.sub_(gn, alpha=lr)                                # z <- z - lr * g
            group["k"] = step
        return loss


def define_optimizer(params, nn_dict, diffusion_process, optim_type: str):
    """define_optimizer(params, nn_dict, diffusion_process, optim_type) of parts/def_for_main.py:119-139."""
    assert optim_type in ["Adam", "AdamW", "RAdamScheduleFree"]
    lr, weight_decay = params["lr"], params["weight_decay"]
    plist = list(nn_dict["egnn"].parameters())
    if params["to_compress_spectrum"]:
        plist += list(nn_dict["spectrum_compressor"].parameters())
    if params["noise_schedule"] == "learned":
        plist += list(diffusion_process.parameters())
    if optim_type == "Adam":
        return torch.optim.Adam(plist, lr=lr, weight_decay=weight_decay)
    if optim_type == "AdamW":
        return torch.optim.AdamW(plist, lr=lr, weight_decay=weight_decay, amsgrad=True)
    return RAdamScheduleFree(plist, lr=lr)
