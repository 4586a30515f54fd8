"""CPU: pin oracle/metrics_ref.py against values produced by the reference's own ssim_loss / psnr_value, and the
qualitative ranges the reference's test_ssim.py checks (identical -> ~1, light noise -> 0.85..0.995, random -> < 0.7)."""
import os

import numpy as np
import torch

from oracle import metrics_ref as M

CASES = ("noisy", "random", "identical", "flat", "small")


def test_metrics_oracle_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for name in CASES:
        x, y = torch.from_numpy(g[f"{name}_x"]), torch.from_numpy(g[f"{name}_y"])
        assert abs(float(M.ssim_loss(x.double(), y.double())) - float(g[f"{name}_ssim_loss"])) < 1e-10, name
        assert abs(float(M.ssim_loss(x, y)) - float(g[f"{name}_ssim_loss_f32"])) < 1e-3, name   # fp32: E[x^2]-mu^2 cancellation noise (flat case)
        assert abs(M.psnr_value(x.double(), y.double()) - float(g[f"{name}_psnr"])) < 1e-9, name


def test_reference_property_ranges():
    gen = torch.Generator().manual_seed(1)
    a = torch.rand(1, 3, 128, 128, generator=gen)
    assert M.ssim_value(a, a) > 0.999 and M.psnr_value(a, a) == 100.0
    assert 0.85 < M.ssim_value(a, a + torch.randn(a.shape, generator=gen) * 0.01) < 0.9999
    assert M.ssim_value(a, torch.rand(a.shape, generator=gen)) < 0.7
