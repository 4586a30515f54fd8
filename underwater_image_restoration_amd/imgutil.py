"""Image conversion / [0, 255] metrics of the IR-SDE test script on the device:
``tensor2img`` and ``calculate_psnr`` with the names and semantics of
/root/reference/legacy/third_party/universal-image-restoration/utils/img_utils.py:136-163,182-189 (called from
config/daclip-sde/test.py:127-169).  The quantisation runs on the GPU and only the uint8 image crosses PCIe."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream


def tensor2img_device(tensor: torch.Tensor, min_max=(0, 1)) -> torch.Tensor:
    """(1,C,H,W) / (C,H,W) / (H,W) fp32 ROCm tensor -> uint8 ROCm tensor (H,W,C) in BGR order, or (H,W)."""
    if not tensor.is_cuda:
        raise RuntimeError("tensor2img (MI355X build) quantises on a ROCm device; there is no CPU fallback")
    t = tensor.squeeze().float()
    if t.dim() == 4:
        raise NotImplementedError("tensor2img: image grids of batches (torchvision make_grid, img_utils.py:145-148) are "
                                  "not part of this build; convert one image at a time")
    if t.dim() not in (2, 3):
        raise TypeError("Only support 4D, 3D and 2D tensor. But received with dimension: {:d}".format(t.dim()))
    t = t.contiguous()
    C_ = 1 if t.dim() == 2 else t.shape[0]
    H, W = t.shape[-2], t.shape[-1]
    out = torch.empty((H, W) if t.dim() == 2 else (H, W, C_), dtype=torch.uint8, device=t.device)
    # 3-D: the reference gathers channels [2, 1, 0] (it assumes 3 of them); reversing is the same thing
    if t.dim() == 3 and C_ != 3:
        raise IndexError("tensor2img: 3-D input must have 3 channels (the reference indexes channels [2, 1, 0])")
    _lib.check(_lib.load().uwir_tensor2img_u8(_p(t), C_, H, W, float(min_max[0]), float(min_max[1]), int(t.dim() == 3),
                                              _p(out), _stream(out)), "tensor2img")
    return out


def tensor2img(tensor: torch.Tensor, out_type=np.uint8, min_max=(0, 1)) -> np.ndarray:
    """Drop-in for img_utils.tensor2img: numpy (H,W,C) BGR / (H,W), uint8."""
    if out_type != np.uint8:
        raise NotImplementedError("tensor2img: only the uint8 output of the reference's callers is built")
    return tensor2img_device(tensor, min_max).cpu().numpy()


def calculate_psnr(img1, img2) -> float:
    """img_utils.calculate_psnr on [0, 255] uint8 images (numpy arrays or ROCm uint8 tensors of one shape)."""
    a = img1 if torch.is_tensor(img1) else torch.from_numpy(np.ascontiguousarray(img1))
    b = img2 if torch.is_tensor(img2) else torch.from_numpy(np.ascontiguousarray(img2))
    if a.dtype != torch.uint8 or b.dtype != torch.uint8 or a.shape != b.shape:
        raise ValueError("calculate_psnr: uint8 images of one shape expected")
    a, b = a.cuda().contiguous(), b.cuda().contiguous()
    acc = torch.empty(1, dtype=torch.int64, device=a.device)
    _lib.check(_lib.load().uwir_sqdiff_u8(_p(a), _p(b), a.numel(), _p(acc), _stream(a)), "sqdiff_u8")
    s = int(acc.item())
    if s == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(s / a.numel()))
