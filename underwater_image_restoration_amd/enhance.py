"""RL adapter operators and state features on the device, with the names, argument
meaning and error behaviour of /root/reference/underwater_ir/rl/uiep_cwgr_ops.py, rl/uiqm_uciqe.py and
rl/infer_adapt.py:52-68.  Images are (C, H, W) fp32 ROCm tensors, as the reference passes them.

Pinned by golden vectors of the reference (tests/golden/enhance.npz): gamma_correction, lf_hf_mix, wavelet_denoise,
guided_filter (one channel: the reference raises on RGB, and so does this), multi_scale_retinex, compute_uicm,
compute_sharpness, state_features.
PARITY UNPINNED (the reference computes them through cv2, which is absent from the build container; restated from
OpenCV's documented arithmetic, tested against oracle/enhance_ref.py): histogram_equalization, clahe_equalization,
compute_uism, compute_uiconm, compute_uciqe, compute_uiqm, build_state, apply_pipeline."""
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


# ---------------------------------------------------------------------------------------------------------------------
# The operators and scores the reference computes through OpenCV.  PARITY UNPINNED: cv2 is not installed in the build
# container, so these follow OpenCV's documented arithmetic (csrc/enhance_cv.hip) and are tested against an independent
# numpy restatement (oracle/enhance_ref.py), not against vectors of the reference.
# ---------------------------------------------------------------------------------------------------------------------
def _rgb(image: torch.Tensor, name: str) -> torch.Tensor:
    x = _image(image, name)
    x = x[0] if x.dim() == 4 and x.shape[0] == 1 else x
    if x.dim() != 3 or x.shape[0] != 3:
        raise ValueError(f"{name}: one (3, H, W) RGB image")      # cv2.cvtColor(..., COLOR_RGB2LAB) needs three channels
    return x


def histogram_equalization(image: torch.Tensor) -> torch.Tensor:
    """uiep_cwgr_ops.py:24-35 on a (C, H, W) image, C <= 3 (cv2.split / merge): per-channel 256-bin equalisation of the
    uint8-quantised image, returned as levels / 255."""
    x = _image(image, "histogram_equalization")
    if x.dim() != 3 or not 1 <= x.shape[0] <= 3:
        raise ValueError("histogram_equalization: one (C, H, W) image with 1 to 3 channels")
    c, h, w = x.shape
    lib = _lib.load()
    ws = _workspace(x.device, int(lib.uwir_cv_workspace_bytes(h, w, 8, 8)))
    out = torch.empty_like(x)
    _lib.check(lib.uwir_histogram_equalization(_p(x), c, h, w, _p(out), _p(ws), ws.numel(), _stream(x)), "histogram_equalization")
    return out


def clahe_equalization(image: torch.Tensor, clip_limit: float = 2.0, tile_grid_size: Tuple[int, int] = (8, 8)) -> torch.Tensor:
    """uiep_cwgr_ops.py:38-44: RGB -> 8-bit Lab, CLAHE on L, back to RGB.  ``tile_grid_size`` is (tiles along x, tiles
    along y) as cv2.createCLAHE takes it."""
    x = _rgb(image, "clahe_equalization")
    _, h, w = x.shape
    tx, ty = int(tile_grid_size[0]), int(tile_grid_size[1])
    lib = _lib.load()
    ws = _workspace(x.device, int(lib.uwir_cv_workspace_bytes(h, w, ty, tx)))
    out = torch.empty_like(x)
    _lib.check(lib.uwir_clahe_equalization(_p(x), h, w, float(clip_limit), ty, tx, _p(out), _p(ws), ws.numel(), _stream(x)),
               "clahe_equalization")
    return out


def _cv_scores(image: torch.Tensor, name: str) -> Tuple[float, float, float]:
    x = _rgb(image, name)
    _, h, w = x.shape
    lib = _lib.load()
    ws = _workspace(x.device, int(lib.uwir_cv_workspace_bytes(h, w, 8, 8)))
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    _lib.check(lib.uwir_cv_scores(_p(x), h, w, _p(out), _p(ws), ws.numel(), _stream(x)), name)
    v = out.tolist()
    return v[0], v[1], v[2]


def compute_uism(image: torch.Tensor) -> float:
    """uiqm_uciqe.py:24-33 applied to ``_image_to_numpy(image)``: mean Sobel gradient magnitude over the three channels."""
    return _cv_scores(image, "compute_uism")[0]


def compute_uiconm(image: torch.Tensor) -> float:
    """uiqm_uciqe.py:36-40: Michelson contrast of the 8-bit L channel."""
    return _cv_scores(image, "compute_uiconm")[1]


def compute_uciqe(image: torch.Tensor) -> float:
    """uiqm_uciqe.py:52-65."""
    return _cv_scores(image, "compute_uciqe")[2]


def compute_uiqm(image: torch.Tensor) -> float:
    """uiqm_uciqe.py:43-49: 0.0282 UICM + 0.2953 UISM + 3.5753 UIConM."""
    uism, uiconm, _ = _cv_scores(image, "compute_uiqm")
    x = image[0] if image.dim() == 4 else image
    return 0.0282 * compute_uicm(x) + 0.2953 * uism + 3.5753 * uiconm


def build_state(image: torch.Tensor, probs: torch.Tensor):
    """infer_adapt.py:62-68: [uiqm, uciqe, mean, std, sharpness] + class probabilities, for a (1, 3, H, W) image."""
    uism, uiconm, uciqe = _cv_scores(image, "build_state")
    uicm, mean, std, sharp = state_features(image)
    uiqm = 0.0282 * uicm + 0.2953 * uism + 3.5753 * uiconm
    return [uiqm, uciqe, mean, std, sharp] + probs.tolist()


def apply_pipeline(image: torch.Tensor, gamma: float, clahe_clip: float, msr_mix: float, lf_hf_alpha: float) -> torch.Tensor:
    """infer_adapt.py:71-79 with the knobs spelled out (EnhancementKnobs.gamma / clahe_clip / msr_mix / lf_hf_alpha) on a
    (1, C, H, W) image.  As in the reference, the guided-filter step takes the whole (C, H, W) image against a (1, 1, k, k)
    box kernel at groups = 1, which raises for C != 1 -- so on RGB frames this raises the reference's RuntimeError (recorded in
    tests/golden/enhance.npz as gf_rgb_raises), and completes for single-channel inputs only up to the Lab step, which
    needs three channels."""
    img = gamma_correction(image.squeeze(0), gamma).unsqueeze(0)
    clahe = clahe_equalization(img.squeeze(0), clip_limit=clahe_clip).unsqueeze(0)
    denoised = wavelet_denoise(clahe.squeeze(0)).unsqueeze(0)
    guided = guided_filter(denoised.squeeze(0), clahe.squeeze(0)).unsqueeze(0)
    retinex = multi_scale_retinex(guided.squeeze(0)).unsqueeze(0)
    mixed = lf_hf_mix(guided, retinex, msr_mix)
    return lf_hf_mix(mixed, retinex, lf_hf_alpha)
