"""Teacher heads on the device: DA-CLIP tokens x prompt embeddings -> per-class masks and
statistics -> degradation code z_d, so teacher -> student runs on the GPU without the .pt / .npy detour.

Same class names, constructor / forward signatures, output dataclasses and errors as
/root/reference/underwater_ir/teacher/{mask_head.py:11-164, differentiable_crf.py:18-80, deg_coder.py:13-59}.  All
arithmetic is one HIP launch (csrc/teacher.hip) per forward; there is no CPU path.

The reference's token-layout heuristic is kept as is (mask_head.py:67-80): a 3-D (B, N, C) tensor with N <= C is read
as (N, B, C), which makes the shipped ViT-B/32 export path fail (49 tokens of width 768).  Feed
4-D (B, C, h, w) tokens -- ``daclip_patch_tokens`` below produces them from the DA-CLIP control tower, projected to
the 512-d embedding space of the prompt embeddings.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .ops import _p, _stream, on_operand_device


@dataclass
class MaskHeadOutput:
    scores: torch.Tensor  # (B, K, H, W)
    probs: torch.Tensor  # (B, K, H, W)
    masks: torch.Tensor  # (B, K, H, W)
    intensity: torch.Tensor  # (B, K, H, W)
    avg_intensity: torch.Tensor  # (B, K)
    global_prob: torch.Tensor  # (B, K)
    confidence: torch.Tensor  # (B, K)
    raw_probs: torch.Tensor  # (B, K, H, W)
    used_crf: bool
    # filled by the same launch (DegradationCoder reads them instead of launching again)
    _logits: Optional[torch.Tensor] = None
    _code: Optional[torch.Tensor] = None
    _confidence_scale: Optional[torch.Tensor] = None
    _eps: float = 1e-5


@dataclass
class DegradationCode:
    code: torch.Tensor  # (B, 2K)
    logits: torch.Tensor  # (B, K)
    avg_intensity: torch.Tensor  # (B, K)
    global_prob: torch.Tensor  # (B, K)
    confidence: torch.Tensor  # (B, K)
    confidence_scale: torch.Tensor  # (B,)
    broadcast: Optional[torch.Tensor] = None  # (B, 2K, H, W) when requested


def _need_cuda(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("teacher heads (MI355X build) run on ROCm tensors only; there is no CPU fallback")


def _ws(dev, B: int, n: int, K: int) -> torch.Tensor:
    need = int(_lib.load().uwir_teacher_heads_workspace_bytes(B, n, K))
    return torch.empty(max(need // 4, 1), dtype=torch.float32, device=dev)


class DifferentiableGuidedCRF(nn.Module):
    """differentiable_crf.py:18-80 (inference only: the device kernel has no backward)."""

    def __init__(self, num_iterations: int = 5, spatial_kernel: int = 5, spatial_sigma: float = 3.0,
                 edge_weight: float = 5.0) -> None:
        super().__init__()
        if spatial_kernel % 2 == 0:
            raise ValueError("spatial_kernel must be odd.")
        self.num_iterations = num_iterations
        self.spatial_kernel = spatial_kernel
        self.spatial_sigma = spatial_sigma
        self.edge_weight = edge_weight
        self.pad = spatial_kernel // 2

    def _fill(self, a: "_lib.TeacherArgs", image: torch.Tensor) -> None:
        a.image = _p(image)
        a.img_c, a.img_h, a.img_w = image.shape[1], image.shape[2], image.shape[3]
        a.use_crf, a.crf_iters, a.crf_ks = 1, self.num_iterations, self.spatial_kernel
        a.crf_sigma, a.crf_edge_weight = float(self.spatial_sigma), float(self.edge_weight)

    @torch.no_grad()
    @on_operand_device
    def forward(self, probs: torch.Tensor, image: torch.Tensor, confidence: Optional[torch.Tensor] = None) -> torch.Tensor:
        _need_cuda(probs, image, confidence)
        if probs.dim() != 4 or image.dim() != 4 or image.shape[0] != probs.shape[0]:
            raise ValueError("probs must be (B, K, H, W) and image (B, C, H', W')")
        B, K, h, w = probs.shape
        pin = probs.contiguous().float()
        img = image.contiguous().float()
        conf = None if confidence is None else confidence.contiguous().float()
        out = torch.empty_like(pin)
        ws = _ws(pin.device, B, h * w, K)
        a = _lib.TeacherArgs()
        a.B, a.h, a.w, a.K, a.C = B, h, w, K, 1
        a.probs = _p(out)
        a.workspace, a.workspace_bytes = _p(ws), ws.numel() * 4
        self._fill(a, img)
        _lib.check(_lib.load().uwir_guided_crf(C.byref(a), _p(pin), _p(conf), _stream(pin)), "guided_crf")
        return out.to(probs.dtype)


class MaskHead(nn.Module):
    """Generate degradation-aware masks and statistics from dense vision-language features (mask_head.py:25-164)."""

    def __init__(self, temperature: float = 10.0, alpha: float = 5.0, default_threshold: float = 0.5,
                 confidence_delta: float = 0.1, morphology_kernel: int = 3, use_crf: bool = False, crf_iterations: int = 5,
                 crf_spatial_kernel: int = 5, crf_spatial_sigma: float = 3.0, crf_edge_weight: float = 5.0) -> None:
        super().__init__()
        self.temperature = temperature
        self.alpha = alpha
        self.default_threshold = default_threshold
        self.confidence_delta = confidence_delta
        self.morphology_kernel = morphology_kernel
        self.use_crf = use_crf
        if use_crf:
            self.crf = DifferentiableGuidedCRF(num_iterations=crf_iterations, spatial_kernel=crf_spatial_kernel,
                                               spatial_sigma=crf_spatial_sigma, edge_weight=crf_edge_weight)
        else:
            self.register_module("crf", None)

    @staticmethod
    def _prepare_image_tokens(tokens: torch.Tensor, grid_size: Optional[Tuple[int, int]]):
        """-> ((B, N, C) view, (h, w)); the layout rules of mask_head.py:59-80, heuristic included."""
        if tokens.dim() == 4:  # (B, C, H, W)
            b, c, h, w = tokens.shape
            return tokens.flatten(2).transpose(1, 2), (h, w)
        if tokens.dim() == 3:
            if not tokens.shape[1] > tokens.shape[2]:  # taken for (N, B, C)
                tokens = tokens.transpose(0, 1)
            if grid_size is None:
                side = int(tokens.shape[1] ** 0.5)
                grid_size = (side, side)
            return tokens, grid_size
        raise ValueError("Unsupported token shape for MaskHead input.")

    @torch.no_grad()
    @on_operand_device
    def forward(self, image_tokens: torch.Tensor, text_embeddings: torch.Tensor,
                grid_size: Optional[Tuple[int, int]] = None, thresholds: Optional[Iterable[float]] = None,
                temperature: Optional[float] = None, alpha: Optional[float] = None,
                confidence_delta: Optional[float] = None, image: Optional[torch.Tensor] = None) -> MaskHeadOutput:
        tokens, (h, w) = self._prepare_image_tokens(image_tokens, grid_size)
        b, n, c = tokens.shape
        if n != h * w:
            raise ValueError(f"Token count {n} does not match grid size {h}x{w}.")
        used_crf = bool(self.use_crf and self.crf is not None)
        if used_crf and image is None:
            raise ValueError("MaskHead with CRF enabled requires the guidance image.")
        _need_cuda(image_tokens, text_embeddings, image)
        k = text_embeddings.shape[0]
        if text_embeddings.dim() != 2 or text_embeddings.shape[1] != c:
            raise RuntimeError(f"token width {c} and text embedding width {tuple(text_embeddings.shape)} do not match")
        dev = tokens.device
        tok = tokens.contiguous().float().reshape(b * n, c)        # the gather of the (B, C, h, w) view is plumbing
        text = text_embeddings.contiguous().float()
        thr = torch.tensor(list(thresholds) if thresholds is not None else [self.default_threshold] * k,
                           dtype=torch.float32).to(dev)
        if thr.numel() != k:
            raise RuntimeError("one threshold per prompt expected")
        maps = torch.empty(5, b, k, h, w, dtype=torch.float32, device=dev)
        stats = torch.empty(4, b, k, dtype=torch.float32, device=dev)
        code = torch.empty(b, 2 * k, dtype=torch.float32, device=dev)
        cscale = torch.empty(b, dtype=torch.float32, device=dev)
        ws = _ws(dev, b, n, k)
        a = _lib.TeacherArgs()
        a.tokens, a.ld_tokens, a.text, a.thresholds = _p(tok), c, _p(text), _p(thr)
        a.B, a.h, a.w, a.C, a.K = b, h, w, c, k
        a.temperature = float(temperature or self.temperature)
        a.alpha = float(alpha or self.alpha)
        a.confidence_delta = float(confidence_delta or self.confidence_delta)
        a.eps = 1e-5
        a.morphology_kernel = max(int(self.morphology_kernel), 1)
        img = None
        if used_crf:
            if image.dim() != 4 or image.shape[0] != b:
                raise ValueError("image must be (B, C, H, W)")
            img = image.contiguous().float()
            self.crf._fill(a, img)
        a.scores, a.probs, a.masks, a.intensity, a.raw_probs = (_p(maps[i]) for i in range(5))
        a.avg_intensity, a.global_prob, a.confidence, a.logits = (_p(stats[i]) for i in range(4))
        a.code, a.confidence_scale = _p(code), _p(cscale)
        a.workspace, a.workspace_bytes = _p(ws), ws.numel() * 4
        _lib.check(_lib.load().uwir_teacher_heads(C.byref(a), _stream(ws)), "teacher_heads")
        return MaskHeadOutput(scores=maps[0], probs=maps[1], masks=maps[2], intensity=maps[3], avg_intensity=stats[0],
                              global_prob=stats[1], confidence=stats[2], raw_probs=maps[4], used_crf=used_crf,
                              _logits=stats[3], _code=code, _confidence_scale=cscale)


class DegradationCoder(nn.Module):
    """Assemble global degradation descriptors from VLM mask statistics (deg_coder.py:24-59)."""

    def __init__(self, eps: float = 1e-5, broadcast: bool = True) -> None:
        super().__init__()
        self.eps = eps
        self.broadcast_enabled = broadcast

    @torch.no_grad()
    def forward(self, mask_output: MaskHeadOutput, spatial_shape: Optional[Tuple[int, int]] = None,
                enable_broadcast: Optional[bool] = None) -> DegradationCode:
        if mask_output._code is None or mask_output._eps != self.eps:
            raise RuntimeError("DegradationCoder (MI355X build) consumes the MaskHeadOutput of this package's MaskHead "
                               "with eps = 1e-5; the code is produced by the same launch")
        code = mask_output._code
        broadcast_tensor = None
        should_broadcast = self.broadcast_enabled if enable_broadcast is None else enable_broadcast
        if should_broadcast and spatial_shape is not None:
            h, w = spatial_shape
            broadcast_tensor = code.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, h, w)   # a view, as in the reference
        return DegradationCode(code=code, logits=mask_output._logits, avg_intensity=mask_output.avg_intensity,
                               global_prob=mask_output.global_prob, confidence=mask_output.confidence,
                               confidence_scale=mask_output._confidence_scale, broadcast=broadcast_tensor)


@torch.no_grad()
def daclip_patch_tokens(clip, image: torch.Tensor) -> torch.Tensor:
    """(B, 3, S, S) CLIP-normalised image -> (B, output_dim, g, g) patch tokens of the DA-CLIP control tower in the
    embedding space of the prompt embeddings: last hidden state (export_pseudolabels.py:274-279), then the tower's own
    ln_post and projection applied to every patch token (what pooling applies to the class token, transformer.py:
    543-555).  This is the projection SURVEY.md section 3.4 asks for: the raw width-768 tokens cannot be compared
    with 512-d text embeddings."""
    _, hiddens = clip.visual_control(image, output_hiddens=True)
    last = hiddens[-1]                                   # (L, B, width), L = 1 + g*g
    return clip.project_tokens("control", last)
