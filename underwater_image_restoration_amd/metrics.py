"""Evaluation metrics of the student's eval loops on the device.

Same names / semantics as /root/reference/underwater_ir/student/train_student.py:70-95: ``ssim_loss`` (11x11 uniform
window), ``ssim_value``, ``psnr_value``; plus ``batch_metrics`` returning per-image values in one pass so that
dataset-scale runs can all-reduce sums instead of gathering images (shard.allreduce_metrics).
"""
from __future__ import annotations

import math

import torch

from . import ops


def _prep(x: torch.Tensor, y: torch.Tensor):
    if not x.is_cuda or not y.is_cuda:
        raise RuntimeError("metrics (MI355X build) run on ROCm tensors only; there is no CPU fallback")
    if x.dim() != 4 or x.shape != y.shape:
        raise ValueError("x and y must be (B, C, H, W) tensors of one shape")
    return x.contiguous().float(), y.contiguous().float()


@torch.no_grad()
def batch_metrics(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """-> [B, 3] fp32 on the device: per image (ssim_loss, mse, psnr_value)."""
    x, y = _prep(x, y)
    return ops.image_metrics(x, y)


def ssim_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mean(1 - ssim_map) over the whole batch (images share one size, so it is the mean of the per-image values)."""
    return batch_metrics(x, y)[:, 0].double().mean().float()


def ssim_value(x: torch.Tensor, y: torch.Tensor) -> float:
    return 1.0 - float(ssim_loss(x, y))


def psnr_value(x: torch.Tensor, y: torch.Tensor) -> float:
    mse = float(batch_metrics(x, y)[:, 1].double().mean())
    return 100.0 if mse <= 1e-10 else 10 * math.log10(1.0 / mse)
