"""Oracle (test infrastructure): torch-CPU restatement of the diffusion process.

Follows /root/reference/diffusion_x_h.py (current x+h process),
/root/reference/E3diffusion_new.py (x-only x_hat formulation) and the schedule
variants of /root/reference/E3diffusion.py.  Random noise is always an explicit
argument here (the reference draws it from torch's global RNG in place,
diffusion_x_h.py:52-53,85-86), so results are deterministic.
"""
from __future__ import annotations

from typing import Optional

import torch


def remove_mean(x: torch.Tensor, batch_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """diffusion_x_h.py:5-14 (== E3diffusion_new.py:5-14).

    With ``batch_index`` the reference works IN PLACE on x; callers that need
    the input preserved must clone first.
    """
    if batch_index is None:
        return x - torch.mean(x, dim=0, keepdim=True)
    num_graph = int(batch_index.max().item()) + 1
    for i in range(num_graph):
        sel = batch_index == i
        x[sel] = x[sel] - torch.mean(x[sel], dim=0, keepdim=True)
    return x


def clip_noise_schedule(alphas2: torch.Tensor, clip_value: float = 0.001) -> torch.Tensor:
    """diffusion_x_h.py:92-97."""
    alphas2 = torch.cat([torch.ones(1), alphas2], dim=0)
    step = torch.clamp(alphas2[1:] / alphas2[:-1], min=clip_value, max=1.0)
    return torch.cumprod(step, dim=0)


def polynomial_schedule(timesteps: int, s: float = 1e-4, power: float = 3.0) -> torch.Tensor:
    """diffusion_x_h.py:99-106.  Returned array is used directly as alpha_t."""
    x = torch.linspace(0, timesteps, timesteps + 1)
    alphas2 = torch.pow(1 - torch.pow(x / timesteps, power), 2)
    alphas2 = clip_noise_schedule(alphas2, 0.001)
    return (1 - 2 * s) * alphas2 + s


def polynomial_schedule_legacy(timesteps: int, s: float = 1e-4, power: float = 3.0) -> torch.Tensor:
    """E3diffusion.py:79-86: steps = T+1, linspace(0, steps, steps)."""
    steps = timesteps + 1
    x = torch.linspace(0, steps, steps)
    alphas2 = torch.pow(1 - torch.pow(x / steps, power), 2)
    alphas2 = clip_noise_schedule(alphas2, 0.001)
    return (1 - 2 * s) * alphas2 + s


def beta_schedule_legacy(initial_beta, final_beta, timesteps, schedule_function="sigmoid"):
    """E3diffusion.py:15-21 -> (beta, alpha, alpha_bar)."""
    if schedule_function == "sigmoid":
        beta = torch.sigmoid(torch.linspace(-6, 6, timesteps + 1)) * (final_beta - initial_beta) + initial_beta
    elif schedule_function == "linear":
        beta = torch.linspace(initial_beta, final_beta, timesteps + 1)
    else:
        raise ValueError(schedule_function)
    alpha = torch.ones(beta.shape) - beta
    return beta, alpha, torch.cumprod(alpha, dim=0)


class DiffusionRef:
    """Restatement of E3DiffusionProcess (diffusion_x_h.py:16-90), 'predefined' schedule,
    or 'learned' with an explicit gamma table (alpha=sqrt(sigmoid(-g)), sigma=sqrt(sigmoid(g)), :40,:46)."""

    def __init__(self, s, power, num_diffusion_timestep, gamma_table: Optional[torch.Tensor] = None):
        self.num_diffusion_timestep = num_diffusion_timestep
        if gamma_table is None:
            self.alpha_schedule = polynomial_schedule(num_diffusion_timestep, s=s, power=power)
            self.sigma_schedule = torch.sqrt(1 - self.alpha_schedule ** 2)
        else:
            g = gamma_table.reshape(-1)
            self.alpha_schedule = torch.sqrt(torch.sigmoid(-g))
            self.sigma_schedule = torch.sqrt(torch.sigmoid(g))

    def alpha(self, t):
        return self.alpha_schedule[t]

    def sigma(self, t):
        return self.sigma_schedule[t]

    def diffuse_zero_to_t(self, z, t, noise, mode="pos"):
        """:51-59 with explicit noise."""
        if mode == "pos":
            noise = remove_mean(noise)
        return self.alpha(t) * z + self.sigma(t) * noise, noise

    def _consts(self, t):
        alpha_t, alpha_s = self.alpha(t), self.alpha(t - 1)
        sq_t, sq_s = 1 - alpha_t ** 2, 1 - alpha_s ** 2
        alpha_ts = alpha_t / alpha_s
        sq_ts = sq_t - torch.pow(alpha_ts, 2) * sq_s
        return alpha_t, alpha_s, sq_t, sq_s, alpha_ts, sq_ts

    def calculate_mu(self, z, eps, t):
        """:61-73."""
        _, _, sq_t, _, alpha_ts, sq_ts = self._consts(t)
        sigma_t = torch.sqrt(sq_t)
        return z / alpha_ts - sq_ts * eps / alpha_ts / sigma_t

    def step_std(self, t):
        _, _, sq_t, sq_s, _, sq_ts = self._consts(t)
        return torch.sqrt(sq_ts * sq_s / sq_t)

    def reverse_diffuse_one_step(self, z, eps, t, noise, mode="pos"):
        """:75-90 with explicit noise."""
        mu = self.calculate_mu(z, eps, t)
        if mode == "pos":
            noise = remove_mean(noise)
        return mu + self.step_std(t) * noise

    # ---- E3diffusion_new.py x_hat formulation (:63-98) ----
    def calculate_mu_xhat(self, pos, eps, t):
        alpha_t, alpha_s, sq_t, sq_s, alpha_ts, sq_ts = self._consts(t)
        x_hat = (pos - self.sigma(t) * eps) / alpha_t
        return alpha_ts * sq_s * pos / sq_t + alpha_s * sq_ts * x_hat / sq_t

    def reverse_from_mu(self, mu, t, noise):
        return mu + self.step_std(t) * remove_mean(noise)

    def step_table(self) -> torch.Tensor:
        """[T+1, 4] per-step constants consumed by the device sampler:
        col0 = 1/alpha_ts, col1 = sigma2_ts/(alpha_ts*sigma_t), col2 = std, col3 = t/T.
        Row 0 holds the final-decode constants of train_per_iretation.py:416-426:
        1/alpha_0, sigma_0/alpha_0, sigma_0/alpha_0, 0."""
        T = self.num_diffusion_timestep
        tab = torch.zeros(T + 1, 4)
        for t in range(1, T + 1):
            _, _, sq_t, _, alpha_ts, sq_ts = self._consts(t)
            tab[t, 0] = 1.0 / alpha_ts
            tab[t, 1] = sq_ts / alpha_ts / torch.sqrt(sq_t)
            tab[t, 2] = self.step_std(t)
            tab[t, 3] = t / T
        a0, s0 = self.alpha(0), self.sigma(0)
        tab[0, 0], tab[0, 1], tab[0, 2], tab[0, 3] = 1.0 / a0, s0 / a0, s0 / a0, 0.0
        return tab


class LegacyRef:
    """Restatement of the oldest process, E3diffusion.py:9-120 (imported by no driver; named by north_star), with explicit
    noise: the beta-schedule class (:15-72) and the polynomial variant (:79-120)."""

    def __init__(self, initial_beta, final_beta, num_diffusion_timestep, schedule_function="sigmoid"):
        self.num_diffusion_timestep = num_diffusion_timestep
        self.beta_schedule, self.alpha_schedule, self.alpha_bar_schedule = beta_schedule_legacy(
            initial_beta, final_beta, num_diffusion_timestep, schedule_function)

    @staticmethod
    def _mu(pos, eps, alpha_t, alpha_s, sq_t, sq_s, x_hat):
        alpha_ts = alpha_t / alpha_s
        sq_ts = sq_t - torch.pow(alpha_ts, 2) * sq_s
        return alpha_ts * sq_s * pos / sq_t + alpha_s * sq_ts * x_hat / sq_t

    @staticmethod
    def _std(alpha_t, alpha_s, sq_t, sq_s):
        alpha_ts = alpha_t / alpha_s
        sq_ts = sq_t - torch.pow(alpha_ts, 2) * sq_s
        return torch.sqrt(sq_ts * sq_s / sq_t)

    def diffuse_zero_to_t(self, pos, t, noise):
        """:23-28 (alpha_bar_t * pos + beta_t * noise, sic)."""
        noise = remove_mean(noise)
        return self.alpha_bar_schedule[t] * pos + self.beta_schedule[t] * noise, noise

    def calculate_mu(self, pos, eps, t):
        """:30-56; x_hat is divided by sqrt(alpha_schedule[t]) (the step's alpha, not alpha_bar -- sic, :39)."""
        ab = self.alpha_bar_schedule
        x_hat = (pos - torch.sqrt(1 - ab[t]) * eps) / torch.sqrt(self.alpha_schedule[t])
        return self._mu(pos, eps, torch.sqrt(ab[t]), torch.sqrt(ab[t - 1]), 1 - ab[t], 1 - ab[t - 1], x_hat)

    def reverse_diffuse_one_step(self, mu, t, noise):
        """:58-72."""
        ab = self.alpha_bar_schedule
        return mu + self._std(torch.sqrt(ab[t]), torch.sqrt(ab[t - 1]), 1 - ab[t], 1 - ab[t - 1]) * remove_mean(noise)

    # polynomial variant: the schedule is rebuilt on every call in the reference (:89, :96, :108)
    def diffuse_to_t(self, pos, t, noise, s=1e-4):
        """:88-94."""
        alpha = polynomial_schedule_legacy(self.num_diffusion_timestep, s=s)
        noise = remove_mean(noise)
        return alpha[t] * pos + torch.sqrt(1 - alpha[t] ** 2) * noise, noise

    def mu_calculate(self, pos, eps, t, s=1e-4):
        """:95-105."""
        alpha = polynomial_schedule_legacy(self.num_diffusion_timestep, s=s)
        sq_t, sq_s = 1 - alpha[t] ** 2, 1 - alpha[t - 1] ** 2
        x_hat = pos / alpha[t] - torch.sqrt(sq_t) / alpha[t] * eps
        return self._mu(pos, eps, alpha[t], alpha[t - 1], sq_t, sq_s, x_hat)

    def reverse_onestep(self, mu, t, noise, s=1e-4):
        """:107-120."""
        alpha = polynomial_schedule_legacy(self.num_diffusion_timestep, s=s)
        return mu + self._std(alpha[t], alpha[t - 1], 1 - alpha[t] ** 2, 1 - alpha[t - 1] ** 2) * remove_mean(noise)
