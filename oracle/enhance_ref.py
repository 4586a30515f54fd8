"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch / numpy) of the cv2-free RL enhancement operators and state
features of the reference.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path never does.

Pinned by tests/golden/enhance.npz, which tests/golden/make_golden.py produced by importing the reference's own
`underwater_ir/rl/uiep_cwgr_ops.py`, `uiqm_uciqe.py` and `infer_adapt.py` (an empty module object stands in for the
absent `cv2` at the import statement only).

NOT restated, "parity unpinned" in this container (they call OpenCV, which the image lacks): `histogram_equalization`
(`cv2.split/merge`), `clahe_equalization`, `compute_uism` (`cv2.Sobel`), `compute_uiconm`, `compute_uciqe` and hence
`compute_uiqm` (`cv2.cvtColor` to 8-bit Lab).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .student_ref import haar_dwt2 as dwt2, haar_idwt2 as idwt2


def gamma_correction(image: torch.Tensor, gamma: float) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:114-116: exponent floored at 1e-3, input clamped to [0, 1]."""
    return image.clamp(0, 1) ** max(gamma, 1e-3)


def lf_hf_mix(low_freq: torch.Tensor, high_freq: torch.Tensor, alpha: float) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:119-120."""
    return (alpha * high_freq + (1 - alpha) * low_freq).clamp(0.0, 1.0)


def wavelet_denoise(image: torch.Tensor, threshold_scale: float = 1.0) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:46-59: one Haar level; per detail band (ALL channels together) a soft threshold
    t = scale * (median|c| / 0.6745)^2 / (sqrt(mean c^2) + 1e-6); torch.median = the LOWER middle element."""
    ll, lh, hl, hh = dwt2(image[None])
    bands = []
    for c in (lh, hl, hh):
        sigma = c.abs().flatten().sort().values[(c.numel() - 1) // 2] / 0.6745
        t = threshold_scale * (sigma ** 2 / (c.pow(2).mean().sqrt() + 1e-6))
        bands.append(torch.sign(c) * torch.relu(c.abs() - t))
    return idwt2(ll, *bands)[0]


def box_mean(x: torch.Tensor, radius: int) -> torch.Tensor:
    """zero-padded (2r+1)^2 box mean of a (1, 1, H, W) tensor (rl/uiep_cwgr_ops.py:62-65 kernel, conv2d padding=r)."""
    k = 2 * radius + 1
    return F.conv2d(x, torch.full((1, 1, k, k), 1.0, dtype=x.dtype) / (k * k), padding=radius)


def guided_filter(input_img: torch.Tensor, guidance: torch.Tensor, radius: int = 4, eps: float = 1e-3) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:68-86.  Defined for ONE channel: the reference convolves with a (1, 1, k, k) kernel at
    groups=1, so a 3-channel image raises RuntimeError there (and here)."""
    p = input_img[None] if input_img.dim() == 3 else input_img
    i = guidance[None] if guidance.dim() == 3 else guidance
    if p.shape[1] != 1 or i.shape[1] != 1:
        raise RuntimeError("guided_filter: the reference's box kernel has one input channel")
    m_i, m_p = box_mean(i, radius), box_mean(p, radius)
    a = (box_mean(i * p, radius) - m_i * m_p) / (box_mean(i * i, radius) - m_i ** 2 + eps)
    b = m_p - a * m_i
    return (box_mean(a, radius) * i + box_mean(b, radius))[0]


def multi_scale_retinex(image: torch.Tensor, sigmas=(15, 80, 250)) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:89-111: x = max(image, 1e-3); mean over sigmas of log x - log(G_sigma * x + 1e-6) with a
    zero-padded, 2-D normalised Gaussian of size 2 ceil(3 sigma) + 1; exp(.) - 1 clamped to [0, 1].  The 2-D kernel
    exp(-(u^2+v^2)/2s^2) / sum factorises exactly into two normalised 1-D kernels, which is how it is applied here
    (the reference's full 2-D convolution needs (6s+1)^2 taps per pixel)."""
    x = image.clamp_min(1e-3)[None]
    c = x.shape[1]
    acc = torch.zeros_like(x)
    for s in sigmas:
        size = int(2 * math.ceil(3 * s) + 1)
        u = torch.arange(size, dtype=torch.float64) - size // 2
        w = torch.exp(-u ** 2 / (2 * s ** 2))
        w = (w / w.sum()).to(x.dtype)
        blur = F.conv2d(x, w.view(1, 1, 1, size).repeat(c, 1, 1, 1), padding=(0, size // 2), groups=c)
        blur = F.conv2d(blur, w.view(1, 1, size, 1).repeat(c, 1, 1, 1), padding=(size // 2, 0), groups=c)
        acc += torch.log(x) - torch.log(blur + 1e-6)
    return (torch.exp(acc / len(sigmas)) - 1.0).clamp(0.0, 1.0)[0]


def to_uint8(image: torch.Tensor) -> np.ndarray:
    """rl/uiqm_uciqe.py:8-10: clamp, HWC, *255, truncate."""
    return (image.clamp(0, 1).permute(1, 2, 0).numpy() * 255).astype(np.uint8)


def compute_uicm(image_u8: np.ndarray) -> float:
    """rl/uiqm_uciqe.py:13-21 on the uint8 HWC image: rg = R - G, yb = (R + G) / 2 - B, population std."""
    r, g, b = (image_u8[:, :, i].astype(np.float64) for i in range(3))
    rg, yb = r - g, 0.5 * (r + g) - b
    return float((-0.0268 * rg.mean() + 0.1586 * rg.std()) - (0.0504 * yb.mean() - 0.164 * yb.std()))


def compute_sharpness(image: torch.Tensor) -> float:
    """rl/infer_adapt.py:52-59: mean Sobel gradient magnitude (zero padding) of the channel-mean of a (1, C, H, W) image."""
    gray = image.mean(dim=1, keepdim=True)
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=image.dtype).view(1, 1, 3, 3)
    gx, gy = F.conv2d(gray, kx, padding=1), F.conv2d(gray, kx.transpose(2, 3), padding=1)
    return float(torch.sqrt(gx ** 2 + gy ** 2).mean())


def histogram_stats(image: torch.Tensor):
    """rl/infer_adapt.py:65-66: mean and UNBIASED std over every element."""
    return float(image.mean()), float(image.std())
