"""``IRSDE`` -- drop-in for the reference sampler with the per-step update fused into one HIP kernel.

Mirrors /root/reference/legacy/third_party/universal-image-restoration/utils/sde_utils.py:80-378 (class ``IRSDE``):
same constructor, same tables (``thetas, sigmas, thetas_cumsum, sigma_bars, dt``), same method names.  The hot
loops ``reverse_sde / reverse_posterior / reverse_ode`` call the score model once per step and then ONE fused
kernel (``uwir_sde_euler_step`` / ``uwir_sde_posterior_step``: 4 reads + 1 write per element instead of ~10
elementwise launches).  Extra, optional argument: ``noise_tape`` ``(T,B,C,H,W)`` of N(0,1) draws (loop order) for
reproducible trajectories; by default noise comes from ``torch.randn_like`` on the device like the reference.
``save_states`` writes the same ``state_{idx}.png`` files as the reference (sde_utils.py:269-275: every T // 100 steps,
the two channel halves of the state side by side) with ``torchvision.utils.save_image``'s defaults restated here
(``state_grid_uint8``: grid of 8 per row, 2 px of zero padding, x * 255 + 0.5 clamped to uint8) and written through PIL;
torchvision is not installed in this image, so the files are "parity unpinned" against it.
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Optional

import torch

from . import ops


def state_grid_uint8(tensor: torch.Tensor, nrow: int = 8, padding: int = 2) -> torch.Tensor:
    """torchvision.utils.make_grid + save_image's quantisation for a (B, C, H, W) batch with default arguments
    (normalize=False, pad_value=0): -> (H', W', 3 or 1) uint8 on the CPU.  Single-channel batches are tripled like
    make_grid does; a single image is returned without a border, also like make_grid."""
    t = tensor.detach().float().cpu()
    if t.dim() != 4:
        raise ValueError("expected a (B, C, H, W) batch")
    if t.shape[1] == 1:
        t = t.expand(-1, 3, -1, -1)
    B, C, H, W = t.shape
    if B == 1:
        grid = t[0]
    else:
        xmaps = min(nrow, B)
        ymaps = int(math.ceil(B / xmaps))
        hh, ww = H + padding, W + padding
        grid = t.new_zeros(C, hh * ymaps + padding, ww * xmaps + padding)
        for k in range(B):
            y, x = k // xmaps, k % xmaps
            grid[:, y * hh + padding:y * hh + padding + H, x * ww + padding:x * ww + padding + W] = t[k]
    return grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)


class IRSDE:
    def __init__(self, max_sigma, T=100, sample_T=-1, schedule="cosine", eps=0.01, device=None):
        self.T = T
        self.device = device
        self.max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma
        self.sample_T = self.T if sample_T < 0 else sample_T
        self.sample_scale = self.T / self.sample_T
        self._initialize(self.max_sigma, self.sample_T, schedule, eps)

    # ---- tables: float32 torch arithmetic in the reference's order, so the scalars are bit-identical
    def _initialize(self, max_sigma, T, schedule, eps=0.01):
        if schedule == "cosine":
            timesteps = T + 2
            x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
            ac = torch.cos(((x / timesteps) + 0.008) / (1 + 0.008) * math.pi * 0.5) ** 2
            ac = ac / ac[0]
            thetas = 1 - ac[1:-1]
        elif schedule == "linear":
            timesteps = T + 1
            scale = 1000 / timesteps
            thetas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float32)
        elif schedule == "constant":
            thetas = torch.ones(T + 1, dtype=torch.float32)
        else:
            raise ValueError(f"Not implemented such schedule yet: {schedule}")
        sigmas = torch.sqrt(max_sigma ** 2 * 2 * thetas)
        thetas_cumsum = torch.cumsum(thetas, dim=0) - thetas[0]
        self.dt = -1 / thetas_cumsum[-1] * math.log(eps)
        sigma_bars = torch.sqrt(max_sigma ** 2 * (1 - torch.exp(-2 * thetas_cumsum * self.dt)))
        self._host = dict(thetas=thetas, sigmas=sigmas, thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars)
        self.thetas = thetas.to(self.device)
        self.sigmas = sigmas.to(self.device)
        self.thetas_cumsum = thetas_cumsum.to(self.device)
        self.sigma_bars = sigma_bars.to(self.device)
        self.mu = 0.0
        self.model = None

    def set_mu(self, mu):
        self.mu = mu

    def set_model(self, model):
        self.model = model

    # ---- small closed-form helpers (scalar-table arithmetic, any device)
    def mu_bar(self, x0, t):
        return self.mu + (x0 - self.mu) * torch.exp(-self.thetas_cumsum[t] * self.dt)

    def sigma_bar(self, t):
        return self.sigma_bars[t]

    def sigma(self, t):
        return self.sigmas[t]

    def theta(self, t):
        return self.thetas[t]

    def drift(self, x, t):
        return self.thetas[t] * (self.mu - x) * self.dt

    def sde_reverse_drift(self, x, score, t):
        return (self.thetas[t] * (self.mu - x) - self.sigmas[t] ** 2 * score) * self.dt

    def ode_reverse_drift(self, x, score, t):
        return (self.thetas[t] * (self.mu - x) - 0.5 * self.sigmas[t] ** 2 * score) * self.dt

    def dispersion(self, x, t):
        return self.sigmas[t] * (torch.randn_like(x) * math.sqrt(self.dt))

    def reverse_optimum_step(self, xt, x0, t):
        term1, term2, _, _ = self._posterior_coeffs(t)
        return float(term1) * (xt - self.mu) + float(term2) * (x0 - self.mu) + self.mu

    def get_score_from_noise(self, noise, t):
        return -noise / self.sigma_bar(t)

    def get_real_noise(self, xt, x0, t):
        return (xt - self.mu_bar(x0, t)) / self.sigma_bar(t)

    def get_real_score(self, xt, x0, t):
        return -(xt - self.mu_bar(x0, t)) / self.sigma_bar(t) ** 2

    def get_init_state_from_noise(self, xt, noise, t):
        A = torch.exp(self.thetas_cumsum[t] * self.dt)
        return (xt - self.mu - self.sigma_bar(t) * noise) * A + self.mu

    def reverse_optimum_std(self, t):
        _, _, std, _ = self._posterior_coeffs(t)
        return std

    def weights(self, t):
        return torch.exp(-self.thetas_cumsum[t] * self.dt)

    def noise_state(self, tensor):
        return tensor + torch.randn_like(tensor) * self.max_sigma

    def _posterior_coeffs(self, t: int):
        h = self._host
        dt = self.dt
        A = torch.exp(-h["thetas"][t] * dt)
        B = torch.exp(-h["thetas_cumsum"][t] * dt)
        C = torch.exp(-h["thetas_cumsum"][t - 1] * dt)
        term1 = A * (1 - C ** 2) / (1 - B ** 2)
        term2 = C * (1 - A ** 2) / (1 - B ** 2)
        A2 = torch.exp(-2 * h["thetas"][t] * dt)
        B2 = torch.exp(-2 * h["thetas_cumsum"][t] * dt)
        C2 = torch.exp(-2 * h["thetas_cumsum"][t - 1] * dt)
        var = (1 - A2) * (1 - C2) / (1 - B2)
        std = (0.5 * torch.log(torch.clamp(var, min=1e-20 * dt))).exp() * self.max_sigma
        expA = torch.exp(h["thetas_cumsum"][t] * dt)
        return term1, term2, std, expA

    # ---- model calls
    def score_fn(self, x, t, scale=1.0, **kwargs):
        return self.get_score_from_noise(self.model(x, self.mu, t * scale, **kwargs), t)

    def noise_fn(self, x, t, scale=1.0, **kwargs):
        return self.model(x, self.mu, t * scale, **kwargs)

    # ---- fused steps (HIP).  ``eps`` = the N(0,1) draw (None -> drawn here; False -> no noise term)
    @staticmethod
    def _prep(*ts):
        out = []
        for t in ts:
            if not t.is_cuda:
                raise RuntimeError("IRSDE (MI355X build): fused steps need ROCm tensors; there is no CPU fallback")
            out.append(t.contiguous().float())
        return out

    def _mu_like(self, x):
        mu = self.mu
        if not torch.is_tensor(mu):
            mu = torch.full_like(x, float(mu))
        return mu.expand_as(x)

    @ops.on_operand_device
    def reverse_sde_step_from_noise(self, x, noise_pred, t: int, eps=None, score_weight: float = 1.0, out=None):
        x, noise_pred, mu = self._prep(x, noise_pred, self._mu_like(x))
        if eps is None:
            eps = torch.randn_like(x)
        elif eps is False:
            eps = None
        else:
            (eps,) = self._prep(eps)
        out = torch.empty_like(x) if out is None else out
        h = self._host
        ops.sde_euler_step(x.view(-1), mu.view(-1), noise_pred.view(-1), None if eps is None else eps.view(-1),
                           out.view(-1), float(h["thetas"][t]), float(h["sigmas"][t]), float(h["sigma_bars"][t]),
                           float(self.dt), score_weight)
        return out

    @ops.on_operand_device
    def reverse_posterior_step(self, xt, noise, t: int, eps=None, out=None):
        xt, noise, mu = self._prep(xt, noise, self._mu_like(xt))
        if eps is None:
            eps = torch.randn_like(xt)
        elif eps is False:
            eps = None
        else:
            (eps,) = self._prep(eps)
        out = torch.empty_like(xt) if out is None else out
        term1, term2, std, expA = self._posterior_coeffs(t)
        ops.sde_posterior_step(xt.view(-1), mu.view(-1), noise.view(-1), None if eps is None else eps.view(-1),
                               out.view(-1), float(self._host["sigma_bars"][t]), float(expA), float(term1), float(term2),
                               float(std))
        return out

    def reverse_sde_step(self, x, score, t: int, eps=None):
        """Reference signature takes the SCORE (sde_utils.py:44-45); noise = -score * sigma_bar."""
        return self.reverse_sde_step_from_noise(x, -score * self.sigma_bar(t), t, eps=eps)

    def reverse_sde_step_mean(self, x, score, t: int):
        return self.reverse_sde_step_from_noise(x, -score * self.sigma_bar(t), t, eps=False)

    def reverse_ode_step(self, x, score, t: int):
        return self.reverse_sde_step_from_noise(x, -score * self.sigma_bar(t), t, eps=False, score_weight=0.5)

    # ---- sampling loops (sde_utils.py:261-313)
    def _save_state(self, x, t: int, save_dir: str) -> None:
        # sde_utils.py:269-275 ("only consider to save 100 images")
        interval = self.T // 100
        if interval == 0:
            raise ZeroDivisionError("save_states needs T >= 100 (the reference computes t % (T // 100))")
        if t % interval:
            return
        from PIL import Image
        os.makedirs(save_dir, exist_ok=True)
        x_l, x_r = x.chunk(2, dim=1)
        if x_l.shape[1] != x_r.shape[1]:
            raise RuntimeError("save_states: the two channel halves of the state differ in size (odd channel count); "
                               "torch.cat fails here in the reference too (sde_utils.py:272)")
        if x_l.shape[1] not in (1, 3):
            raise ValueError(f"save_states: {x_l.shape[1]}-channel halves cannot be written as an image (PIL, like "
                             "torchvision.utils.save_image, takes 1 or 3 channels)")
        img = state_grid_uint8(torch.cat([x_l, x_r], dim=3))
        arr = img.numpy()
        Image.fromarray(arr if arr.shape[2] == 3 else arr[:, :, 0]).save(f"{save_dir}/state_{t // interval}.png")

    def _loop(self, xt, T, mode, noise_tape, kwargs, save_dir=None):
        T = self.sample_T if T < 0 else T
        x = xt.clone().contiguous().float()
        # a ConditionalUNet of this package derives everything that depends on (t, contexts) alone for the whole loop in
        # one batched pass (unet.conditioning_schedule); any other callable is stepped as the reference does.  Contract:
        # the context tensors and the model's weights are not modified while the loop runs (the schedule notices an
        # in-place edit of a context through tensor._version and then recomputes per step like the reference)
        sched = getattr(self.model, "conditioning_schedule", None)
        ok = sched is not None and x.is_cuda and set(kwargs) <= {"text_context", "image_context"}
        ctx = (sched([t * self.sample_scale for t in range(1, T + 1)], x.shape[0], device=x.device, **kwargs) if ok
               else contextlib.nullcontext())
        with ctx:
            for i, t in enumerate(reversed(range(1, T + 1))):
                pred = self.noise_fn(x, t, self.sample_scale, **kwargs)
                eps = None if noise_tape is None else noise_tape[i]
                if mode == "posterior":
                    x = self.reverse_posterior_step(x, pred, t, eps=eps, out=x)
                elif mode == "sde":
                    x = self.reverse_sde_step_from_noise(x, pred, t, eps=eps, out=x)
                else:
                    x = self.reverse_sde_step_from_noise(x, pred, t, eps=False, score_weight=0.5, out=x)
                if save_dir is not None:
                    self._save_state(x, t, save_dir)
        return x

    def reverse_sde(self, xt, T=-1, save_states=False, save_dir="sde_state", noise_tape=None, **kwargs):
        return self._loop(xt, T, "sde", noise_tape, kwargs, save_dir if save_states else None)

    def reverse_posterior(self, xt, T=-1, save_states=False, save_dir="posterior_state", noise_tape=None, **kwargs):
        return self._loop(xt, T, "posterior", noise_tape, kwargs, save_dir if save_states else None)

    def reverse_ode(self, xt, T=-1, save_states=False, save_dir="ode_state", **kwargs):
        return self._loop(xt, T, "ode", None, kwargs, save_dir if save_states else None)
