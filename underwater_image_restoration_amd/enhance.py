"""RL adapter operators and state features on the device, with the names, argument
meaning and error behaviour of /root/reference/underwater_ir/rl/uiep_cwgr_ops.py, rl/uiqm_uciqe.py and
rl/infer_adapt.py:52-68.  Images are (C, H, W) fp32 ROCm tensors, as the reference passes them.

Built: gamma_correction, lf_hf_mix, wavelet_denoise, guided_filter (one channel: the reference raises on RGB, and so
does this), multi_scale_retinex, compute_uicm, compute_sharpness, state_features.
Not built (they call OpenCV, absent from this image, so any restatement would be parity-unpinned): the functions
listed in ``OPENCV_ONLY`` raise NotImplementedError."""
from __future__ import annotations

import ctypes
from typing import Sequence, Tuple

import torch

from . import _lib
from .ops import _p, _stream

_WS = {}


def _image(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} (MI355X build) runs on a ROCm device; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected a float32 tensor")
    return t.contiguous()


def _workspace(dev, nbytes: int) -> torch.Tensor:
    t = _WS.get(dev)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[dev] = t
    return t


def gamma_correction(image: torch.Tensor, gamma: float) -> torch.Tensor:
    """uiep_cwgr_ops.py:114-116."""
    x = _image(image, "gamma_correction")
    out = torch.empty_like(x)
    _lib.check(_lib.load().uwir_gamma_correction(_p(x), _p(out), x.numel(), float(gamma), _stream(x)), "gamma_correction")
    return out


def lf_hf_mix(low_freq: torch.Tensor, high_freq: torch.Tensor, alpha: float) -> torch.Tensor:
    """uiep_cwgr_ops.py:119-120 (same-shape operands)."""
    lo, hi = _image(low_freq, "lf_hf_mix"), _image(high_freq, "lf_hf_mix")
    if lo.shape != hi.shape:
        raise RuntimeError(f"lf_hf_mix: shapes differ: {tuple(lo.shape)} vs {tuple(hi.shape)}")
    out = torch.empty_like(lo)
    _lib.check(_lib.load().uwir_lf_hf_mix(_p(lo), _p(hi), float(alpha), _p(out), lo.numel(), _stream(lo)), "lf_hf_mix")
    return out


def wavelet_denoise(image: torch.Tensor, threshold_scale: float = 1.0) -> torch.Tensor:
    """uiep_cwgr_ops.py:46-59 on a (C, H, W) image with even H, W."""
    x = _image(image, "wavelet_denoise")
    if x.dim() != 3:
        raise ValueError("Input to dwt2 must be 4D (B, C, H, W).")   # the reference unsqueezes a 3-D image and dwt2 checks
    c, h, w = x.shape
    lib = _lib.load()
    ws = _workspace(x.device, int(lib.uwir_wavelet_denoise_workspace_bytes(c, h, w)))
    out = torch.empty_like(x)
    _lib.check(lib.uwir_wavelet_denoise(_p(x), c, h, w, float(threshold_scale), _p(out), _p(ws), ws.numel(), _stream(x)),
               "wavelet_denoise")
    return out


def guided_filter(input_img: torch.Tensor, guidance: torch.Tensor, radius: int = 4, eps: float = 1e-3) -> torch.Tensor:
    """uiep_cwgr_ops.py:68-86; (1, H, W) or (1, 1, H, W) operands, returns (1, H, W)."""
    p, g = _image(input_img, "guided_filter"), _image(guidance, "guided_filter")
    p = p[None] if p.dim() == 3 else p
    g = g[None] if g.dim() == 3 else g
    if p.dim() != 4 or g.dim() != 4 or p.shape[1] != 1 or g.shape[1] != 1:
        # F.conv2d with the reference's (1, 1, k, k) kernel at groups=1
        raise RuntimeError(f"Given groups=1, weight of size [1, 1, {2 * radius + 1}, {2 * radius + 1}], expected input"
                           f"{list(p.shape)} to have 1 channels, but got {p.shape[1]} channels instead")
    if p.shape != g.shape or p.shape[0] != 1:
        raise RuntimeError("guided_filter: one (1, H, W) image and a guidance of the same shape")
    h, w = p.shape[-2:]
    lib = _lib.load()
    ws = _workspace(p.device, int(lib.uwir_guided_filter_workspace_bytes(h, w)))
    out = torch.empty(1, h, w, dtype=torch.float32, device=p.device)
    _lib.check(lib.uwir_guided_filter(_p(p), _p(g), h, w, int(radius), float(eps), _p(out), _p(ws), ws.numel(), _stream(p)),
               "guided_filter")
    return out


def multi_scale_retinex(image: torch.Tensor, sigmas: Sequence[float] = (15, 80, 250)) -> torch.Tensor:
    """uiep_cwgr_ops.py:89-111 on a (C, H, W) image."""
    x = _image(image, "multi_scale_retinex")
    x = x[0] if x.dim() == 4 and x.shape[0] == 1 else x
    if x.dim() != 3:
        raise RuntimeError("multi_scale_retinex: one (C, H, W) image")
    c, h, w = x.shape
    lib = _lib.load()
    ws = _workspace(x.device, int(lib.uwir_multi_scale_retinex_workspace_bytes(c, h, w)))
    sg = (ctypes.c_double * len(sigmas))(*[float(s) for s in sigmas])
    out = torch.empty_like(x)
    _lib.check(lib.uwir_multi_scale_retinex(_p(x), c, h, w, sg, len(sigmas), _p(out), _p(ws), ws.numel(), _stream(x)),
               "multi_scale_retinex")
    return out


def state_features(image: torch.Tensor) -> Tuple[float, float, float, float]:
    """(UICM, mean, unbiased std, sharpness) of a (C, H, W) or (1, C, H, W) image: the OpenCV-free entries of
    infer_adapt.build_state (:62-68).  One launch pair, four doubles cross PCIe."""
    x = _image(image, "state_features")
    x = x[0] if x.dim() == 4 and x.shape[0] == 1 else x
    if x.dim() != 3:
        raise RuntimeError("state_features: one (C, H, W) image")
    c, h, w = x.shape
    lib = _lib.load()
    ws = _workspace(x.device, int(lib.uwir_enhance_state_workspace_bytes()))
    out = torch.empty(4, dtype=torch.float64, device=x.device)
    _lib.check(lib.uwir_enhance_state(_p(x), c, h, w, _p(out), _p(ws), ws.numel(), _stream(x)), "enhance_state")
    v = out.tolist()
    return v[0], v[1], v[2], v[3]


def compute_uicm(image: torch.Tensor) -> float:
    """uiqm_uciqe.py:13-21 applied to ``_image_to_numpy(image)`` (:8-10): takes the (3, H, W) float image."""
    if image.shape[-3] != 3:
        raise IndexError("compute_uicm: the reference indexes channels 0..2")
    return state_features(image)[0]


def compute_sharpness(image: torch.Tensor) -> float:
    """infer_adapt.py:52-59 on a (1, C, H, W) image."""
    return state_features(image)[3]


OPENCV_ONLY = ("histogram_equalization", "clahe_equalization", "compute_uism", "compute_uiconm", "compute_uiqm", "compute_uciqe")


def __getattr__(name: str):
    if name in OPENCV_ONLY:
        def missing(*_a, **_k):
            raise NotImplementedError(f"{name}: the reference computes this with OpenCV (CLAHE / 8-bit Lab / cv2.Sobel), "
                                      "which this build has no pinned restatement of (DESIGN.md section 7)")
        return missing
    raise AttributeError(name)
