"""Diffusion process with the reference's interface (diffusion_x_h.py:16-106; x-only variant of
E3diffusion_new.py:16-115).  The schedule is evaluated by the library's host routine
(schedule_table_build) and the per-step arithmetic runs in HIP kernels; Gaussian noise is drawn with
torch's generator on the tensor's device, as the reference does (bit-parity of samples
across devices is impossible, only eps-parity and statistics)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .snr import GammaNetwork


def _graph_ptr_from_batch(batch_index: torch.Tensor, num_graphs: Optional[int] = None):
    b = batch_index.long()
    if num_graphs is None:   # the reference's own form (diffusion_x_h.py:10): one host sync
        if b.numel() > 1 and bool((b[1:] < b[:-1]).any()):
            raise ValueError("batch_index must be sorted (PyG collate order)")
        nb = int(b.max().item()) + 1
    else:                    # collated batches know their graph count: no host sync
        nb = int(num_graphs)
    cnt = torch.zeros(nb, dtype=torch.long, device=b.device).index_add_(0, b, torch.ones_like(b))
    ptr = torch.zeros(nb + 1, dtype=torch.int32, device=b.device)
    ptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    return ptr, nb


def remove_mean(x: torch.Tensor, batch_index: Optional[torch.Tensor] = None, num_graphs: Optional[int] = None) -> torch.Tensor:
    """remove_mean(x, batch_index=None) (diffusion_x_h.py:5-14).  Like the reference, the
    batch_index form updates ``x`` in place and returns it; the global form returns a new tensor.
    ``num_graphs`` (extension) skips the host sync that finds the graph count."""
    if not x.is_cuda:
        raise RuntimeError("remove_mean needs a CUDA(ROCm) tensor; there is no CPU fallback")
    xc = x.detach().to(torch.float32).contiguous()
    out = torch.empty_like(xc)
    n, d = xc.shape
    if batch_index is None:
        _lib.check(_lib.lib().egnn_remove_mean(_lib.stream_ptr(), n, d, None, 0, _lib.ptr(xc), _lib.ptr(out)))
        return out
    ptr, nb = _graph_ptr_from_batch(batch_index.to(x.device), num_graphs)
    _lib.check(_lib.lib().egnn_remove_mean(_lib.stream_ptr(), n, d, _lib.ptr(ptr), nb, _lib.ptr(xc), _lib.ptr(out)))
    x.copy_(out)
    return x


def build_schedule(T: int, s: float, power: float):
    """(alpha [T+1], sigma [T+1], step table [T+1,4]) as CPU fp32 tensors (polynomial_schedule +
    clip_noise_schedule, diffusion_x_h.py:92-106)."""
    a = np.empty(T + 1, dtype=np.float32)
    sg = np.empty(T + 1, dtype=np.float32)
    tab = np.empty((T + 1) * 4, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.lib().schedule_table_build(int(T), float(s), float(power), a.ctypes.data_as(fp),
                                               sg.ctypes.data_as(fp), tab.ctypes.data_as(fp)))
    return torch.from_numpy(a), torch.from_numpy(sg), torch.from_numpy(tab).view(T + 1, 4)


def table_from_alpha(alpha: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    a = alpha.detach().cpu().float().contiguous().numpy()
    sg = sigma.detach().cpu().float().contiguous().numpy()
    T = a.shape[0] - 1
    tab = np.empty((T + 1) * 4, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.lib().schedule_table_from_alpha(T, a.ctypes.data_as(fp), sg.ctypes.data_as(fp), tab.ctypes.data_as(fp)))
    return torch.from_numpy(tab).view(T + 1, 4)


class E3DiffusionProcess(nn.Module):
    """E3DiffusionProcess(s, power, num_diffusion_timestep, noise_schedule='predefined')
    (diffusion_x_h.py:16-30)."""

    def __init__(self, s, power, num_diffusion_timestep: int, noise_schedule: str = "predefined"):
        super().__init__()
        self.noise_schedule = noise_schedule
        self.num_diffusion_timestep = num_diffusion_timestep
        if noise_schedule == "predefined":
            self.noise_precision, self.power = s, power
            self.t = torch.linspace(0, num_diffusion_timestep, num_diffusion_timestep + 1)
            self.alpha_schedule, self.sigma_schedule, self._table = build_schedule(num_diffusion_timestep, s, power)
        elif noise_schedule == "learned":
            self.gamma = GammaNetwork()
            self.t = torch.linspace(0, 1, num_diffusion_timestep + 1).view(num_diffusion_timestep + 1, 1)
            self._table, self._gamma_sig = None, None
        else:
            raise ValueError(noise_schedule)
        self._table_dev = {}

    # -- schedule ------------------------------------------------------------------------------------
    def gamma_schedule(self):
        return self.gamma(self.t.to(self.gamma.gamma_0.device))

    def _learned_tables(self):
        """alpha = sqrt(sigmoid(-gamma)), sigma = sqrt(sigmoid(gamma)) over the whole grid (:40,:46),
        tabulated once per parameter version instead of once per call."""
        sig = tuple((p.data_ptr(), p._version) for p in self.gamma.parameters())
        if sig != self._gamma_sig:
            with torch.no_grad():
                g = self.gamma_schedule().reshape(-1).float().cpu()
            self.alpha_schedule = torch.sqrt(torch.sigmoid(-g))
            self.sigma_schedule = torch.sqrt(torch.sigmoid(g))
            self._table = table_from_alpha(self.alpha_schedule, self.sigma_schedule)
            self._gamma_sig, self._table_dev = sig, {}

    def alpha(self, t: int):
        if self.noise_schedule == "learned":
            self._learned_tables()
        return self.alpha_schedule[t]

    def sigma(self, t: int):
        if self.noise_schedule == "learned":
            self._learned_tables()
        return self.sigma_schedule[t]

    def alpha_sigma_tables(self, device, with_grad: bool = False):
        """(alpha [T+1], sigma [T+1]) on ``device`` for indexed gathers by a vector of times.  For the learned
        schedule with ``with_grad`` the whole grid is evaluated through GammaNetwork under autograd
        (alpha = sqrt(sigmoid(-gamma)), sigma = sqrt(sigmoid(gamma)), diffusion_x_h.py:36-46), so that what is built
        from the gathered values is differentiable w.r.t. the schedule parameters as in the reference."""
        if self.noise_schedule == "learned" and with_grad and any(p.requires_grad for p in self.gamma.parameters()):
            g = self.gamma_schedule().reshape(-1).float().to(device)
            return torch.sqrt(torch.sigmoid(-g)), torch.sqrt(torch.sigmoid(g))
        if self.noise_schedule == "learned":
            self._learned_tables()
        key = "as:" + str(device)
        if key not in self._table_dev:
            self._table_dev[key] = (self.alpha_schedule.to(device).float().contiguous(),
                                    self.sigma_schedule.to(device).float().contiguous())
        return self._table_dev[key]

    def step_table(self, device=None) -> torch.Tensor:
        """[T+1, 4] per-step constants {1/alpha_ts, sigma2_ts/(alpha_ts sigma_t), std, t/T}."""
        if self.noise_schedule == "learned":
            self._learned_tables()
        if device is None:
            return self._table
        key = str(device)
        if key not in self._table_dev:
            self._table_dev[key] = self._table.to(device).contiguous()
        return self._table_dev[key]

    # -- forward / reverse steps ---------------------------------------------------------------------
    @staticmethod
    def _prep(z):
        if not z.is_cuda:
            raise RuntimeError("E3DiffusionProcess steps need CUDA(ROCm) tensors; there is no CPU fallback")
        return z.detach().to(torch.float32).contiguous()

    def diffuse_zero_to_t(self, z: torch.Tensor, t: int, mode="pos", noise: Optional[torch.Tensor] = None):
        """z_t = alpha_t z + sigma_t eps, eps ~ N(0,I) mean-removed iff mode == 'pos' (:51-59); ``noise``
        (extension, as in reverse_diffuse_one_step) replaces the N(0,I) draw BEFORE the mean removal."""
        zc = self._prep(z)
        noise = torch.zeros_like(zc).normal_(mean=0, std=1) if noise is None else self._prep(noise).clone()
        if mode == "pos":
            noise = remove_mean(noise)
        out = torch.empty_like(zc)
        n, d = zc.shape
        _lib.check(_lib.lib().ddpm_reverse_step(_lib.stream_ptr(), n, d, 0, None, 0, float(self.alpha(t)), 0.0,
                                                float(self.sigma(t)), _lib.ptr(zc), d, _lib.ptr(zc), _lib.ptr(noise),
                                                _lib.ptr(out), d))
        return out, noise

    def calculate_mu(self, z: torch.Tensor, epsilon: torch.Tensor, t: int):
        """mu = z/alpha_ts - sigma2_ts eps/(alpha_ts sigma_t) (:61-73)."""
        zc, ec = self._prep(z), self._prep(epsilon)
        c = self.step_table()[t]
        out = torch.empty_like(zc)
        n, d = zc.shape
        _lib.check(_lib.lib().ddpm_reverse_step(_lib.stream_ptr(), n, d, 0, None, 0, float(c[0]), float(c[1]), 0.0,
                                                _lib.ptr(zc), d, _lib.ptr(ec), _lib.ptr(ec), _lib.ptr(out), d))
        return out

    def reverse_diffuse_one_step(self, z, epsilon: torch.Tensor, t: int, mode="pos", noise: Optional[torch.Tensor] = None):
        """z_{t-1} = mu + std * noise (:75-90); ``noise`` (extension) overrides the RNG draw."""
        zc, ec = self._prep(z), self._prep(epsilon)
        if noise is None:
            noise = torch.zeros_like(zc).normal_(mean=0, std=1)
        nz = self._prep(noise)
        c = self.step_table()[t]
        out = torch.empty_like(zc)
        n, d = zc.shape
        _lib.check(_lib.lib().ddpm_reverse_step(_lib.stream_ptr(), n, d, 1 if mode == "pos" else 0, None, 0,
                                                float(c[0]), float(c[1]), float(c[2]), _lib.ptr(zc), d, _lib.ptr(ec),
                                                _lib.ptr(nz), _lib.ptr(out), d))
        return out

    # -- reference helpers kept for API parity -------------------------------------------------------
    def clip_noise_schedule(self, alphas2, clip_value=0.001):
        alphas2 = torch.cat([torch.ones(1), alphas2], dim=0)
        step = torch.clamp(alphas2[1:] / alphas2[:-1], min=clip_value, max=1.0)
        return torch.cumprod(step, dim=0)

    def polynomial_schedule(self, timesteps: int, s=1e-4, power=3.0):
        return build_schedule(timesteps, s, power)[0]


class E3DiffusionProcessXOnly(E3DiffusionProcess):
    """Interface of the x-only process of E3diffusion_new.py:16-98 (used by the reference's train.py /
    test.py): diffuse_zero_to_t(pos, t), calculate_mu(pos, eps, t) in the x_hat form (algebraically the
    same mu as diffusion_x_h), reverse_diffuse_one_step(mu, t)."""

    def diffuse_zero_to_t(self, pos, t, noise: Optional[torch.Tensor] = None):  # noqa: D102
        return super().diffuse_zero_to_t(pos, t, mode="pos", noise=noise)

    def reverse_diffuse_one_step(self, mu, t, noise: Optional[torch.Tensor] = None):  # noqa: D102
        mc = self._prep(mu)
        if noise is None:
            noise = torch.zeros_like(mc).normal_(mean=0, std=1)
        nz = self._prep(noise)
        out = torch.empty_like(mc)
        n, d = mc.shape
        _lib.check(_lib.lib().ddpm_reverse_step(_lib.stream_ptr(), n, d, 1, None, 0, 1.0, 0.0,
                                                float(self.step_table()[t][2]), _lib.ptr(mc), d, _lib.ptr(mc),
                                                _lib.ptr(nz), _lib.ptr(out), d))
        return out


class E3DiffusionProcessLegacy:
    """The oldest process of the reference (E3diffusion.py:9-120, named by north_star; imported by no driver): the
    beta-schedule class -- ``E3DiffusionProcess(initial_beta, final_beta, num_diffusion_timestep, schedule_function)`` with
    ``diffuse_zero_to_t(pos, t)`` (:23-28: alpha_bar_t pos + beta_t noise, sic), ``calculate_mu(pos, eps, t)`` (:30-56, x_hat
    divided by sqrt(alpha_t) of the step, sic) and ``reverse_diffuse_one_step(mu, t)`` (:58-72) -- and its second, polynomial
    variant ``diffuse_to_t / mu_calculate / reverse_onestep`` (:88-120; schedule with steps = T + 1, :79-86).

    The [T+1] schedules are built on the host with the reference's own torch expressions; every step is a linear
    combination c0 z - c1 eps + c2 remove_mean(noise) and runs in the ``ddpm_reverse_step`` kernel.  ``noise`` (extension,
    as in E3DiffusionProcess) replaces the N(0, I) draw before the mean removal."""

    def __init__(self, initial_beta, final_beta, num_diffusion_timestep: int, schedule_function="sigmoid"):
        self.initial_beta, self.final_beta = initial_beta, final_beta
        self.schedule_function = schedule_function
        self.num_diffusion_timestep = num_diffusion_timestep
        if schedule_function == "sigmoid":      # :15-17
            self.beta_schedule = torch.sigmoid(torch.linspace(-6, 6, num_diffusion_timestep + 1))
            self.beta_schedule = self.beta_schedule * (final_beta - initial_beta) + initial_beta
        elif schedule_function == "linear":     # :18-19
            self.beta_schedule = torch.linspace(initial_beta, final_beta, num_diffusion_timestep + 1)
        else:
            raise ValueError(schedule_function)
        self.alpha_schedule = torch.ones(self.beta_schedule.shape) - self.beta_schedule      # :20
        self.alpha_bar_schedule = torch.cumprod(self.alpha_schedule, dim=0)                  # :21
        self._poly = {}

    # -- the three kernels' constants -----------------------------------------------------------------
    @staticmethod
    def _mix(z, eps, noise, c0, c1, c2, remove):
        zc, ec = E3DiffusionProcess._prep(z), E3DiffusionProcess._prep(eps)
        if noise is None:
            noise = torch.zeros_like(zc).normal_(mean=0, std=1)
        nz = E3DiffusionProcess._prep(noise)
        out = torch.empty_like(zc)
        n, d = zc.shape
        _lib.check(_lib.lib().ddpm_reverse_step(_lib.stream_ptr(), n, d, 1 if remove else 0, None, 0, float(c0), float(c1),
                                                float(c2), _lib.ptr(zc), d, _lib.ptr(ec), _lib.ptr(nz), _lib.ptr(out), d))
        return out

    @staticmethod
    def _step_consts(alpha_t, alpha_s, sq_t, sq_s, xhat_div, xhat_sig):
        """mu = alpha_ts sq_s z / sq_t + alpha_s sq_ts x_hat / sq_t with x_hat = (z - xhat_sig eps) / xhat_div, as fp32
        scalars in the reference's operation order -> (c0, c1, std)"""
        alpha_ts = alpha_t / alpha_s
        sq_ts = sq_t - torch.pow(alpha_ts, 2) * sq_s
        a, b = alpha_ts * sq_s / sq_t, alpha_s * sq_ts / sq_t
        return a + b / xhat_div, b * xhat_sig / xhat_div, torch.sqrt(sq_ts * sq_s / sq_t)

    def _beta_consts(self, t):
        ab = self.alpha_bar_schedule
        return self._step_consts(torch.sqrt(ab[t]), torch.sqrt(ab[t - 1]), 1 - ab[t], 1 - ab[t - 1],
                                 torch.sqrt(self.alpha_schedule[t]), torch.sqrt(1 - ab[t]))

    # -- beta-schedule class (:23-72) ------------------------------------------------------------------
    def diffuse_zero_to_t(self, pos, t: int, noise: Optional[torch.Tensor] = None):
        pc = E3DiffusionProcess._prep(pos)
        noise = torch.zeros_like(pc).normal_(mean=0, std=1) if noise is None else E3DiffusionProcess._prep(noise).clone()
        noise = remove_mean(noise)
        return self._mix(pc, pc, noise, self.alpha_bar_schedule[t], 0.0, self.beta_schedule[t], False), noise

    def calculate_mu(self, pos, epsilon, t: int):
        c0, c1, _ = self._beta_consts(t)
        return self._mix(pos, epsilon, epsilon, c0, c1, 0.0, False)

    def reverse_diffuse_one_step(self, mu, t: int, noise: Optional[torch.Tensor] = None):
        return self._mix(mu, mu, noise, 1.0, 0.0, self._beta_consts(t)[2], True)

    # -- polynomial variant (:79-120) ------------------------------------------------------------------
    def clip_noise_schedule(self, alphas2, clip_value=0.001):
        alphas2 = torch.cat([torch.ones(1), alphas2], dim=0)
        return torch.cumprod(torch.clamp(alphas2[1:] / alphas2[:-1], min=clip_value, max=1.0), dim=0)

    def polynomial_schedule(self, timesteps: int, s=1e-4, power=3.0):
        steps = timesteps + 1                                            # :80-81 (sic: not the grid of diffusion_x_h.py)
        x = torch.linspace(0, steps, steps)
        alphas2 = self.clip_noise_schedule(torch.pow(1 - torch.pow(x / steps, power), 2), clip_value=0.001)
        return (1 - 2 * s) * alphas2 + s

    def _poly_consts(self, t, s):
        if s not in self._poly:      # the reference rebuilds the schedule on every call (:89, :96, :108)
            self._poly[s] = self.polynomial_schedule(self.num_diffusion_timestep, s=s)
        alpha = self._poly[s]
        sq_t, sq_s = 1 - alpha[t] ** 2, 1 - alpha[t - 1] ** 2
        return alpha, self._step_consts(alpha[t], alpha[t - 1], sq_t, sq_s, alpha[t], torch.sqrt(sq_t))

    def diffuse_to_t(self, pos, t: int, s=1e-4, noise: Optional[torch.Tensor] = None):
        alpha, _ = self._poly_consts(max(t, 1), s)
        pc = E3DiffusionProcess._prep(pos)
        noise = torch.zeros_like(pc).normal_(mean=0, std=1) if noise is None else E3DiffusionProcess._prep(noise).clone()
        noise = remove_mean(noise)
        return self._mix(pc, pc, noise, alpha[t], 0.0, torch.sqrt(1 - alpha[t] ** 2), False), noise

    def mu_calculate(self, pos, epsilon, t: int, s=1e-4):
        c0, c1, _ = self._poly_consts(t, s)[1]
        return self._mix(pos, epsilon, epsilon, c0, c1, 0.0, False)

    def reverse_onestep(self, mu, t: int, s=1e-4, noise: Optional[torch.Tensor] = None):
        return self._mix(mu, mu, noise, 1.0, 0.0, self._poly_consts(t, s)[1][2], True)
