"""CPU oracle for the IR-SDE sampler (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates /root/reference/legacy/third_party/universal-image-restoration/utils/sde_utils.py:
schedules + tables (:91-154), reverse Euler-Maruyama step (:44-45,177-187), posterior step (:205-247) and the
three sampling loops (:261-313), on torch CPU tensors.  Noise is passed in explicitly (``noise_tape[i]`` is the
N(0,1) draw of loop iteration i) so trajectories are reproducible on any device.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch


def theta_schedule(schedule: str, T: int) -> torch.Tensor:
    if schedule == "cosine":  # :112-123
        timesteps = T + 2
        x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
        ac = torch.cos(((x / timesteps) + 0.008) / 1.008 * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        return 1 - ac[1:-1]
    if schedule == "linear":  # :100-110
        timesteps = T + 1
        scale = 1000 / timesteps
        return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float32)
    if schedule == "constant":  # :92-98
        return torch.ones(T + 1, dtype=torch.float32)
    raise ValueError(schedule)


class SdeTables:
    def __init__(self, max_sigma: float, T: int = 100, sample_T: int = -1, schedule: str = "cosine", eps: float = 0.01):
        self.T = T
        self.max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma  # :86
        self.sample_T = T if sample_T < 0 else sample_T
        self.sample_scale = self.T / self.sample_T
        thetas = theta_schedule(schedule, self.sample_T)
        self.thetas = thetas
        self.sigmas = torch.sqrt(self.max_sigma ** 2 * 2 * thetas)              # :128-129
        self.thetas_cumsum = torch.cumsum(thetas, dim=0) - thetas[0]            # :144
        self.dt = -1 / self.thetas_cumsum[-1] * math.log(eps)                   # :145 (overwrites 1/T)
        self.sigma_bars = torch.sqrt(self.max_sigma ** 2 * (1 - torch.exp(-2 * self.thetas_cumsum * self.dt)))  # :131-132

    # ---- single steps
    def reverse_sde_step(self, x, mu, noise_pred, t: int, eps_noise: Optional[torch.Tensor], score_weight: float = 1.0):
        score = -noise_pred / self.sigma_bars[t]                                                     # :186-187
        drift = (self.thetas[t] * (mu - x) - score_weight * self.sigmas[t] ** 2 * score) * self.dt  # :177-181
        out = x - drift
        if eps_noise is not None:
            out = out - self.sigmas[t] * (eps_noise * math.sqrt(self.dt))                            # :183-184
        return out

    def posterior_coeffs(self, t: int):
        A = torch.exp(-self.thetas[t] * self.dt)
        B = torch.exp(-self.thetas_cumsum[t] * self.dt)
        C = torch.exp(-self.thetas_cumsum[t - 1] * self.dt)
        term1 = A * (1 - C ** 2) / (1 - B ** 2)                                  # :205-214
        term2 = C * (1 - A ** 2) / (1 - B ** 2)
        A2 = torch.exp(-2 * self.thetas[t] * self.dt)
        B2 = torch.exp(-2 * self.thetas_cumsum[t] * self.dt)
        C2 = torch.exp(-2 * self.thetas_cumsum[t - 1] * self.dt)
        var = (1 - A2) * (1 - C2) / (1 - B2)                                     # :216-225
        std = (0.5 * torch.log(torch.clamp(var, min=1e-20 * self.dt))).exp() * self.max_sigma
        expA = torch.exp(self.thetas_cumsum[t] * self.dt)                        # :245-247
        return term1, term2, std, expA

    def reverse_posterior_step(self, x, mu, noise_pred, t: int, eps_noise: Optional[torch.Tensor]):
        term1, term2, std, expA = self.posterior_coeffs(t)
        x0 = (x - mu - self.sigma_bars[t] * noise_pred) * expA + mu
        mean = term1 * (x - mu) + term2 * (x0 - mu) + mu
        return mean if eps_noise is None else mean + std * eps_noise

    # ---- loops: model(x, mu, t * sample_scale) -> predicted noise
    def sample(self, model: Callable, xt: torch.Tensor, mu: torch.Tensor, mode: str, noise_tape: Optional[torch.Tensor],
               T: int = -1) -> torch.Tensor:
        T = self.sample_T if T < 0 else T
        x = xt.clone()
        for i, t in enumerate(reversed(range(1, T + 1))):
            pred = model(x, mu, t * self.sample_scale)
            e = None if noise_tape is None else noise_tape[i]
            if mode == "posterior":
                x = self.reverse_posterior_step(x, mu, pred, t, e)
            elif mode == "sde":
                x = self.reverse_sde_step(x, mu, pred, t, e)
            elif mode == "ode":
                x = self.reverse_sde_step(x, mu, pred, t, None, score_weight=0.5)
            else:
                raise ValueError(mode)
        return x
