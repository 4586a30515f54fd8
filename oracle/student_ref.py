"""CPU oracle: functional restatement of the student net ``NAFNetWFIGate``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Layout is the reference's
own (NCHW, torch CPU); every function works on a flat ``state_dict`` (``sd``)
and a key prefix so no ``nn.Module`` of the reference is needed at run time.

Reference files (relative to /root/reference/underwater_ir/student/naf_unet_wfi/):
  model.py, wtb.py, sffb.py, cfc.py, wfi_gate.py, dwt_fft.py, utils.py, deg_head.py
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- #
# dwt_fft.py:13-58  orthonormal Haar analysis / synthesis
# --------------------------------------------------------------------------- #
def _haar_coeff(dtype: torch.dtype) -> torch.Tensor:
    # dwt_fft.py:10-21: the 2x2 filters are outer products of [s, s] / [s, -s]
    # with s = 2**-0.5 rounded to the working dtype, so the tap is s*s (not 0.5).
    s = torch.tensor(2 ** -0.5, dtype=dtype)
    return s * s


def haar_dwt2(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """dwt_fft.py:33-39.  (B,C,2h,2w) -> ll, lh, hl, hh each (B,C,h,w).

    2x2 block [[a,b],[c,d]]: first filter index is the row (dwt_fft.py:17-20):
    lh = row-low x col-high -> (a-b+c-d)/2, hl -> (a+b-c-d)/2.  Odd trailing
    rows / columns are dropped exactly like a stride-2 valid conv does.
    """
    if x.dim() != 4:
        raise ValueError("Input to dwt2 must be 4D (B, C, H, W).")
    h2, w2 = (x.shape[2] // 2) * 2, (x.shape[3] // 2) * 2
    a = x[:, :, 0:h2:2, 0:w2:2]
    b = x[:, :, 0:h2:2, 1:w2:2]
    c = x[:, :, 1:h2:2, 0:w2:2]
    d = x[:, :, 1:h2:2, 1:w2:2]
    k = _haar_coeff(x.dtype)
    ll = k * a + k * b + k * c + k * d
    lh = k * a - k * b + k * c - k * d
    hl = k * a + k * b - k * c - k * d
    hh = k * a - k * b - k * c + k * d
    return ll, lh, hl, hh


def haar_idwt2(ll: torch.Tensor, lh: torch.Tensor, hl: torch.Tensor, hh: torch.Tensor) -> torch.Tensor:
    """dwt_fft.py:42-58.  Four (B,C,h,w) sub-bands -> (B,C,2h,2w)."""
    if ll.dim() != 4:
        raise ValueError("Subbands must be 4D tensors.")
    k = _haar_coeff(ll.dtype)
    bsz, ch, h, w = ll.shape
    out = ll.new_empty(bsz, ch, 2 * h, 2 * w)
    out[:, :, 0::2, 0::2] = k * ll + k * lh + k * hl + k * hh
    out[:, :, 0::2, 1::2] = k * ll - k * lh + k * hl - k * hh
    out[:, :, 1::2, 0::2] = k * ll + k * lh - k * hl - k * hh
    out[:, :, 1::2, 1::2] = k * ll - k * lh - k * hl + k * hh
    return out


# --------------------------------------------------------------------------- #
# utils.py:10-36   LayerNorm2d / AdaLayerNorm2d  (statistics over H,W per (b,c))
# --------------------------------------------------------------------------- #
def layernorm2d(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)  # biased, utils.py:19
    y = (x - mean) / torch.sqrt(var + eps)
    return y * sd[p + "weight"] + sd[p + "bias"]


def ada_layernorm2d(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
    """utils.py:24-36: [scale|shift] = Linear(cond); out = LN(x)*(1+scale)+shift."""
    ss = F.linear(cond, sd[p + "linear.weight"], sd[p + "linear.bias"])
    c = ss.shape[1] // 2
    scale = ss[:, :c, None, None]
    shift = ss[:, c:, None, None]
    return layernorm2d(sd, p + "norm.", x) * (1 + scale) + shift


def conv1x1(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.conv2d(x, sd[p + "weight"], sd.get(p + "bias"))


def depthwise(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    w = sd[p + "weight"]
    return F.conv2d(x, w, sd.get(p + "bias"), padding=w.shape[-1] // 2, groups=w.shape[0])


# --------------------------------------------------------------------------- #
# wtb.py:13-124   WideTransformerBlock (+ ChannelAttention)
# --------------------------------------------------------------------------- #
def _global_attention(q: torch.Tensor, k: torch.Tensor, vs: List[torch.Tensor], chunk: int = 1024) -> List[torch.Tensor]:
    """softmax(q^T k / sqrt(hd)) applied to each v.  q,k,v: (B,heads,hd,n).

    The reference chunks over queries purely to bound memory (wtb.py:104-112);
    any chunk size gives the same numbers, so the oracle picks its own.
    """
    hd = q.shape[2]
    n = q.shape[3]
    outs = [[] for _ in vs]
    for i in range(0, n, chunk):
        s = torch.matmul(q[..., i:i + chunk].transpose(-2, -1), k) / math.sqrt(hd)
        a = torch.softmax(s, dim=-1)
        for j, v in enumerate(vs):
            outs[j].append(torch.matmul(a, v.transpose(-2, -1)).transpose(-2, -1))
    return [torch.cat(o, dim=-1) for o in outs]


def channel_attention(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """wtb.py:13-28."""
    pooled = x.mean(dim=(2, 3))
    h = F.relu(F.linear(pooled, sd[p + "net.0.weight"], sd[p + "net.0.bias"]))
    w = torch.sigmoid(F.linear(h, sd[p + "net.2.weight"], sd[p + "net.2.bias"]))
    return x * w[:, :, None, None]


def wide_transformer_block(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor, num_heads: int,
                           taps: Optional[dict] = None) -> torch.Tensor:
    """wtb.py:83-124."""
    b, c, h, w = x.shape
    hd = c // num_heads
    xn = ada_layernorm2d(sd, p + "norm1.", x, cond)
    qkv_base = conv1x1(sd, p + "qkv.", xn)
    multi = torch.cat([depthwise(sd, p + f"dw_branches.{i}.", qkv_base) for i in range(3)], dim=1)
    qkv = conv1x1(sd, p + "dw_merge.", multi)
    q, k, v = torch.chunk(qkv, 3, dim=1)
    q = q.reshape(b, num_heads, hd, h * w)
    k = k.reshape(b, num_heads, hd, h * w)
    v = v.reshape(b, num_heads, hd, h * w)
    (att,) = _global_attention(q, k, [v])
    att = att.reshape(b, c, h, w)
    x1 = x + conv1x1(sd, p + "proj.", att) + channel_attention(sd, p + "channel_attn.", xn)
    xn2 = ada_layernorm2d(sd, p + "norm2.", x1, cond)
    ff = conv1x1(sd, p + "ffn.2.", F.gelu(conv1x1(sd, p + "ffn.0.", xn2)))
    if taps is not None:
        taps.update(xn=xn, qkv_base=qkv_base, qkv=qkv, att=att, x1=x1)
    return x1 + ff


# --------------------------------------------------------------------------- #
# sffb.py:12-55 + dwt_fft.py:61-72   SpatialFrequencyFusionBlock
# --------------------------------------------------------------------------- #
def spatial_frequency_fusion(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor,
                             taps: Optional[dict] = None) -> torch.Tensor:
    xn = ada_layernorm2d(sd, p + "norm.", x, cond)
    branches = []
    for i in range(3):
        d = depthwise(sd, p + f"spatial_convs.{i}.depthwise.", xn)
        branches.append(conv1x1(sd, p + f"spatial_convs.{i}.pointwise.", d))
    spatial = conv1x1(sd, p + "merge.", torch.cat(branches, dim=1))
    spec = torch.fft.fft2(spatial, norm="ortho")
    im = spec.imag.clone()
    # Self-conjugate bins (ky in {0,h/2}, kx in {0,w/2}) of a real signal are exactly real.
    # torch's CPU FFT returns +0 there for most sizes (all powers of two), but for some
    # (e.g. 24x16) it leaves +-1e-17 rounding noise whose SIGN flips atan2 between +pi
    # and -pi, i.e. the reference is a coin flip there.  The oracle pins the exact value.
    hh_, ww_ = spec.shape[-2:]
    for ky in ({0, hh_ // 2} if hh_ % 2 == 0 else {0}):
        for kx in ({0, ww_ // 2} if ww_ % 2 == 0 else {0}):
            im[..., ky, kx] = 0.0
    amp = torch.complex(spec.real, im).abs()
    phase = torch.atan2(im, spec.real)
    amp2 = conv1x1(sd, p + "amp_mod.", amp)
    ph2 = conv1x1(sd, p + "phase_mod.", phase)
    freq = torch.fft.ifft2(torch.complex(amp2 * torch.cos(ph2), amp2 * torch.sin(ph2)), norm="ortho").real
    if taps is not None:
        taps.update(spatial=spatial, amp=amp, phase=phase, freq=freq)
    return conv1x1(sd, p + "reproj.", spatial + freq)


# --------------------------------------------------------------------------- #
# cfc.py:10-39   CrossFrequencyConditioner: one attention map, two value sets
# --------------------------------------------------------------------------- #
def cross_frequency_conditioner(sd: SD, p: str, hf: torch.Tensor, lf: torch.Tensor, num_heads: int = 4):
    b, c, h, w = hf.shape
    hd = c // num_heads
    shp = (b, num_heads, hd, h * w)
    q = conv1x1(sd, p + "q_conv.", hf).reshape(shp)
    k = conv1x1(sd, p + "k_conv.", lf).reshape(shp)
    vt = conv1x1(sd, p + "vt_conv.", hf).reshape(shp)
    vf = conv1x1(sd, p + "vf_conv.", lf).reshape(shp)
    t_out, f_out = _global_attention(q, k, [vt, vf])
    return t_out.reshape(b, c, h, w), f_out.reshape(b, c, h, w)


# --------------------------------------------------------------------------- #
# wfi_gate.py:11-45
# --------------------------------------------------------------------------- #
def wfi_gate(sd: SD, p: str, hf: torch.Tensor, lf: torch.Tensor, cond: torch.Tensor,
             masks: Optional[torch.Tensor]):
    cp = F.linear(cond, sd[p + "cond_proj.weight"], sd[p + "cond_proj.bias"])
    feats = [hf, lf, cp[:, :, None, None].expand(-1, -1, hf.shape[2], hf.shape[3])]
    if masks is not None:
        if masks.shape[2:] != hf.shape[2:]:
            masks = F.interpolate(masks, size=hf.shape[-2:], mode="bilinear", align_corners=False)
        feats.append(masks)
    z = torch.cat(feats, dim=1)
    hid = F.relu(conv1x1(sd, p + "net.0.", z))
    alpha = torch.sigmoid(conv1x1(sd, p + "net.2.", hid))
    return alpha * hf + (1 - alpha) * lf, alpha


# --------------------------------------------------------------------------- #
# model.py:17-41   WFIBlock
# --------------------------------------------------------------------------- #
def wfi_block(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor],
              taps: Optional[dict] = None):
    ll, lh, hl, hh = haar_dwt2(x)
    t_w: Optional[dict] = {} if taps is not None else None
    t_s: Optional[dict] = {} if taps is not None else None
    hf = wide_transformer_block(sd, p + "hf_block.", torch.cat([lh, hl, hh], dim=1), cond,
                                num_heads=12, taps=t_w)
    lh, hl, hh = torch.chunk(hf, 3, dim=1)
    lf = spatial_frequency_fusion(sd, p + "lf_block.", ll, cond, taps=t_s)
    t_out, f_out = cross_frequency_conditioner(sd, p + "cfc.", lh + hl + hh, lf, num_heads=4)
    merged, alpha = wfi_gate(sd, p + "gate.", t_out, f_out, cond, masks)
    out = haar_idwt2(merged, lh, hl, hh)
    if taps is not None:
        taps.update(ll=ll, hf=hf, lf=lf, t_out=t_out, f_out=f_out, merged=merged, alpha=alpha, out=out,
                    wtb=t_w, sffb=t_s)
    return out, alpha


def degradation_head(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """deg_head.py:7-22."""
    pooled = x.mean(dim=(2, 3), keepdim=True)
    h = F.gelu(conv1x1(sd, p + "net.0.", pooled))
    return conv1x1(sd, p + "net.2.", h).flatten(1)


# --------------------------------------------------------------------------- #
# model.py:64-162   NAFNetWFIGate.forward
# --------------------------------------------------------------------------- #
def student_forward(sd: SD, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor] = None,
                    num_levels: int = 3, blocks_per_level: int = 2, taps: Optional[dict] = None):
    """Returns (restored, alpha_maps, deg_logits); ``taps`` (if a dict) collects
    per-block intermediates and the pre-clamp head output."""
    if cond is None:
        raise ValueError("Condition vector z_d must be provided.")
    residual = x
    x = F.conv2d(x, sd["stem.weight"], sd["stem.bias"], padding=1)
    if taps is not None:
        taps["stem"] = x
    skips: List[torch.Tensor] = []
    alphas: List[torch.Tensor] = []

    def run_stage(prefix: str, x: torch.Tensor) -> torch.Tensor:
        for bi in range(blocks_per_level):
            bt = {} if taps is not None else None
            x, a = wfi_block(sd, f"{prefix}.blocks.{bi}.", x, cond, masks, taps=bt)
            alphas.append(a)
            if taps is not None:
                taps[f"{prefix}.blocks.{bi}"] = bt
        return x

    for lvl in range(num_levels):
        x = run_stage(f"encoder_stages.{lvl}", x)
        if lvl < num_levels - 1:
            skips.append(x)
            x = F.conv2d(x, sd[f"downsamples.{lvl}.weight"], sd[f"downsamples.{lvl}.bias"], stride=2)
    logits = degradation_head(sd, "degradation_head.", x) if "degradation_head.net.0.weight" in sd else None
    for di in range(num_levels - 1):
        skip = skips[num_levels - 2 - di]
        x = F.conv_transpose2d(x, sd[f"upsamples.{di}.weight"], sd[f"upsamples.{di}.bias"], stride=2)
        if x.shape[-2:] != skip.shape[-2:]:  # model.py:150-151
            x = F.interpolate(x, size=skip.shape[-2:], mode="bilinear", align_corners=False)
        x = torch.cat([x, skip], dim=1)
        x = conv1x1(sd, f"skip_fusions.{di}.", x)
        x = run_stage(f"decoder_stages.{di}", x)
    x = layernorm2d(sd, "final_norm.", x)
    if taps is not None:
        taps["final_norm"] = x
    pre = F.conv2d(x, sd["head.weight"], sd["head.bias"], padding=1)
    if taps is not None:
        taps["pre_clamp"] = pre + residual
    return torch.clamp(pre + residual, 0, 1), alphas, logits
