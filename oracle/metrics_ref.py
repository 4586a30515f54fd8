"""CPU oracle for the evaluation metrics (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates /root/reference/underwater_ir/student/train_student.py:70-95 (``ssim_loss`` with the 11x11 uniform window,
``ssim_value``, ``psnr_value``; identical copies live in /root/reference/test_ssim.py:10-38, which is what the golden
vectors in tests/golden/metrics.npz were generated from -- the package itself does not import here, torchvision is
absent).  Written as explicit box sums instead of avg_pool2d.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _box(t: torch.Tensor, k: int) -> torch.Tensor:
    """k x k box mean with zero padding that is COUNTED in the divisor (avg_pool2d default)."""
    p = k // 2
    tp = F.pad(t, (p, p, p, p))
    c = tp.cumsum(dim=-1)
    c = F.pad(c, (1, 0))
    h = c[..., k:] - c[..., :-k]
    c = h.cumsum(dim=-2)
    c = F.pad(c, (0, 0, 1, 0))
    return (c[..., k:, :] - c[..., :-k, :]) / (k * k)


def ssim_loss(x: torch.Tensor, y: torch.Tensor, window_size: int = 11, c1: float = 0.01 ** 2, c2: float = 0.03 ** 2):
    mu_x, mu_y = _box(x, window_size), _box(y, window_size)
    sx = (_box(x * x, window_size) - mu_x ** 2).clamp(min=0)        # :73-79
    sy = (_box(y * y, window_size) - mu_y ** 2).clamp(min=0)
    sxy = _box(x * y, window_size) - mu_x * mu_y
    num = (2 * mu_x * mu_y + c1) * (2 * sxy + c2)
    den = (mu_x ** 2 + mu_y ** 2 + c1) * (sx + sy + c2)
    return (1 - num / (den + 1e-8)).mean()                          # :81-84


def ssim_value(x, y) -> float:
    return 1.0 - float(ssim_loss(x, y))


def psnr_value(x, y) -> float:
    mse = float(torch.mean((x - y) ** 2))                           # :91-95
    return 100.0 if mse <= 1e-10 else 10 * math.log10(1.0 / mse)
