"""Deterministic synthetic weights and inputs (no checkpoints / datasets exist offline).

``student_param_shapes`` is the weight ABI of the student net: the exact
``state_dict`` key names and shapes of the reference ``NAFNetWFIGate``
(/root/reference/underwater_ir/student/naf_unet_wfi/model.py:64-122 and the
sub-module constructors it calls).  tests/golden/make_golden.py asserts it
equal to the reference's own ``state_dict()`` listing.
"""
from __future__ import annotations

import hashlib
import re
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch


def _wfi_block_shapes(p: str, c: int, cond_dim: int, num_masks: int, out: "OrderedDict[str, Tuple[int, ...]]") -> None:
    cw = 3 * c
    h = p + "hf_block."
    for nm in ("norm1", "norm2"):
        out[h + nm + ".norm.weight"] = (1, cw, 1, 1)
        out[h + nm + ".norm.bias"] = (1, cw, 1, 1)
        out[h + nm + ".linear.weight"] = (2 * cw, cond_dim)
        out[h + nm + ".linear.bias"] = (2 * cw,)
    out[h + "qkv.weight"] = (3 * cw, cw, 1, 1)
    out[h + "qkv.bias"] = (3 * cw,)
    for i, k in enumerate((3, 5, 7)):
        out[h + f"dw_branches.{i}.weight"] = (3 * cw, 1, k, k)
        out[h + f"dw_branches.{i}.bias"] = (3 * cw,)
    out[h + "dw_merge.weight"] = (3 * cw, 9 * cw, 1, 1)
    out[h + "dw_merge.bias"] = (3 * cw,)
    out[h + "proj.weight"] = (cw, cw, 1, 1)
    out[h + "proj.bias"] = (cw,)
    hid = max(cw // 8, 4)
    out[h + "channel_attn.net.0.weight"] = (hid, cw)
    out[h + "channel_attn.net.0.bias"] = (hid,)
    out[h + "channel_attn.net.2.weight"] = (cw, hid)
    out[h + "channel_attn.net.2.bias"] = (cw,)
    out[h + "ffn.0.weight"] = (2 * cw, cw, 1, 1)
    out[h + "ffn.0.bias"] = (2 * cw,)
    out[h + "ffn.2.weight"] = (cw, 2 * cw, 1, 1)
    out[h + "ffn.2.bias"] = (cw,)
    l = p + "lf_block."
    out[l + "norm.norm.weight"] = (1, c, 1, 1)
    out[l + "norm.norm.bias"] = (1, c, 1, 1)
    out[l + "norm.linear.weight"] = (2 * c, cond_dim)
    out[l + "norm.linear.bias"] = (2 * c,)
    for i, k in enumerate((3, 5, 7)):
        out[l + f"spatial_convs.{i}.depthwise.weight"] = (c, 1, k, k)
        out[l + f"spatial_convs.{i}.depthwise.bias"] = (c,)
        out[l + f"spatial_convs.{i}.pointwise.weight"] = (c, c, 1, 1)
        out[l + f"spatial_convs.{i}.pointwise.bias"] = (c,)
    out[l + "merge.weight"] = (c, 3 * c, 1, 1)
    out[l + "merge.bias"] = (c,)
    for nm in ("amp_mod", "phase_mod", "reproj"):
        out[l + nm + ".weight"] = (c, c, 1, 1)
        out[l + nm + ".bias"] = (c,)
    f = p + "cfc."
    for nm in ("q_conv", "k_conv", "vt_conv", "vf_conv"):
        out[f + nm + ".weight"] = (c, c, 1, 1)
        out[f + nm + ".bias"] = (c,)
    g = p + "gate."
    out[g + "cond_proj.weight"] = (c, cond_dim)
    out[g + "cond_proj.bias"] = (c,)
    out[g + "net.0.weight"] = (64, 3 * c + max(num_masks, 0), 1, 1)
    out[g + "net.0.bias"] = (64,)
    out[g + "net.2.weight"] = (c, 64, 1, 1)
    out[g + "net.2.bias"] = (c,)


def student_param_shapes(in_channels: int = 3, out_channels: int = 3, base_channels: int = 64, num_levels: int = 3,
                         blocks_per_level: int = 2, cond_dim: int = 16, num_masks: int = 0,
                         num_degradation_types: Optional[int] = None) -> "OrderedDict[str, Tuple[int, ...]]":
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch = [base_channels * 2 ** i for i in range(num_levels)]
    out["stem.weight"] = (base_channels, in_channels, 3, 3)
    out["stem.bias"] = (base_channels,)
    out["final_norm.weight"] = (1, base_channels, 1, 1)
    out["final_norm.bias"] = (1, base_channels, 1, 1)
    out["head.weight"] = (out_channels, base_channels, 3, 3)
    out["head.bias"] = (out_channels,)
    for lvl in range(num_levels):
        for b in range(blocks_per_level):
            _wfi_block_shapes(f"encoder_stages.{lvl}.blocks.{b}.", ch[lvl], cond_dim, num_masks, out)
    for lvl in range(num_levels - 1):
        out[f"downsamples.{lvl}.weight"] = (ch[lvl + 1], ch[lvl], 2, 2)
        out[f"downsamples.{lvl}.bias"] = (ch[lvl + 1],)
    rev = list(reversed(range(num_levels - 1)))
    for di, lvl in enumerate(rev):
        out[f"upsamples.{di}.weight"] = (ch[lvl + 1], ch[lvl], 2, 2)  # ConvTranspose2d: (in, out, kh, kw)
        out[f"upsamples.{di}.bias"] = (ch[lvl],)
    for di, lvl in enumerate(rev):
        out[f"skip_fusions.{di}.weight"] = (ch[lvl], 2 * ch[lvl], 1, 1)
        out[f"skip_fusions.{di}.bias"] = (ch[lvl],)
    for di, lvl in enumerate(rev):
        for b in range(blocks_per_level):
            _wfi_block_shapes(f"decoder_stages.{di}.blocks.{b}.", ch[lvl], cond_dim, num_masks, out)
    if num_degradation_types is not None:
        out["degradation_head.net.0.weight"] = (256, ch[-1], 1, 1)
        out["degradation_head.net.0.bias"] = (256,)
        out["degradation_head.net.2.weight"] = (num_degradation_types, 256, 1, 1)
        out["degradation_head.net.2.bias"] = (num_degradation_types,)
    return out


def _vit_tower_shapes(p: str, tp: str, image_size: int, patch: int, width: int, layers: int, out_dim: int,
                      out: "OrderedDict[str, Tuple[int, ...]]") -> None:
    """One open_clip ``VisionTransformer`` (L/open_clip/transformer.py:376-442); ``tp`` = prefix of its resblocks."""
    grid = image_size // patch
    out[p + "class_embedding"] = (width,)
    out[p + "positional_embedding"] = (grid * grid + 1, width)
    out[p + "proj"] = (width, out_dim)
    out[p + "conv1.weight"] = (width, 3, patch, patch)
    out[p + "ln_pre.weight"] = (width,)
    out[p + "ln_pre.bias"] = (width,)
    for i in range(layers):
        r = f"{tp}resblocks.{i}."
        out[r + "ln_1.weight"] = (width,)
        out[r + "ln_1.bias"] = (width,)
        out[r + "attn.in_proj_weight"] = (3 * width, width)
        out[r + "attn.in_proj_bias"] = (3 * width,)
        out[r + "attn.out_proj.weight"] = (width, width)
        out[r + "attn.out_proj.bias"] = (width,)
        out[r + "ln_2.weight"] = (width,)
        out[r + "ln_2.bias"] = (width,)
        out[r + "mlp.c_fc.weight"] = (4 * width, width)
        out[r + "mlp.c_fc.bias"] = (4 * width,)
        out[r + "mlp.c_proj.weight"] = (width, 4 * width)
        out[r + "mlp.c_proj.bias"] = (width,)
    out[p + "ln_post.weight"] = (width,)
    out[p + "ln_post.bias"] = (width,)


def daclip_vision_param_shapes(image_size: int = 224, patch_size: int = 32, width: int = 768, layers: int = 12,
                               output_dim: int = 512) -> "OrderedDict[str, Tuple[int, ...]]":
    """Weight ABI of the vision side of ``DaCLIP`` (L/open_clip/daclip_model.py:17-24): ``visual.*`` is the CLIP
    ViT, ``visual_control.*`` its copy whose transformer is wrapped in ``ControlTransformer``
    (L/open_clip/transformer.py:288-306: ``transformer.transformer.resblocks.*`` + ``transformer.zero_modules.*``).
    Defaults = model_configs/daclip_ViT-B-32.json."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _vit_tower_shapes("visual.", "visual.transformer.", image_size, patch_size, width, layers, output_dim, out)
    _vit_tower_shapes("visual_control.", "visual_control.transformer.transformer.", image_size, patch_size, width, layers,
                      output_dim, out)
    for i in range(layers):
        out[f"visual_control.transformer.zero_modules.{i}.weight"] = (width, width)
        out[f"visual_control.transformer.zero_modules.{i}.bias"] = (width,)
    return out


def unet_param_shapes(in_nc: int = 3, out_nc: int = 3, nf: int = 64, ch_mult=(1, 2, 4, 8), context_dim: int = 512,
                      use_degra_context: bool = True, use_image_context: bool = True
                      ) -> "OrderedDict[str, Tuple[int, ...]]":
    """Weight ABI of the IR-SDE ``ConditionalUNet`` (U/models/modules/DenoisingUNet_arch.py:22-109 and the
    constructors in module_util.py / attention.py it calls).  Defaults = U/options/test.yml:30-39."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    td = nf * 4
    ctx = context_dim if context_dim is not None else -1
    if ctx > 0 and use_degra_context:
        out["prompt"] = (1, td)
    out["init_conv.weight"] = (nf, 2 * in_nc, 7, 7)
    out["time_mlp.1.weight"], out["time_mlp.1.bias"] = (td, nf), (td,)
    out["time_mlp.3.weight"], out["time_mlp.3.bias"] = (td, td), (td,)
    if ctx > 0 and use_degra_context:
        out["text_mlp.0.weight"], out["text_mlp.0.bias"] = (td, ctx), (td,)
        out["text_mlp.2.weight"], out["text_mlp.2.bias"] = (td, td), (td,)
        out["prompt_mlp.weight"], out["prompt_mlp.bias"] = (td, td), (td,)

    def resblock(p: str, din: int, dout: int) -> None:
        out[p + "mlp.1.weight"], out[p + "mlp.1.bias"] = (2 * dout, td), (2 * dout,)
        out[p + "block1.proj.weight"] = (dout, din, 3, 3)
        out[p + "block2.proj.weight"] = (dout, dout, 3, 3)
        if din != dout:
            out[p + "res_conv.weight"] = (dout, din, 1, 1)

    def attn(p: str, dim: int, spatial: bool) -> None:
        if not spatial:                                   # Residual(PreNorm(dim, LinearAttention(dim)))
            out[p + "fn.fn.to_qkv.weight"] = (384, dim, 1, 1)
            out[p + "fn.fn.to_out.0.weight"], out[p + "fn.fn.to_out.0.bias"] = (dim, 128, 1, 1), (dim,)
            out[p + "fn.fn.to_out.1.g"] = (1, dim, 1, 1)
        else:                                             # Residual(PreNorm(dim, SpatialTransformer(dim, dim//32, 32)))
            f = p + "fn.fn."
            out[f + "norm.weight"], out[f + "norm.bias"] = (dim,), (dim,)
            out[f + "proj_in.weight"], out[f + "proj_in.bias"] = (dim, dim, 1, 1), (dim,)
            t = f + "transformer_blocks.0."
            for nm in ("to_q", "to_k", "to_v"):
                out[t + f"attn1.{nm}.weight"] = (dim, dim)
            out[t + "attn1.to_out.0.weight"], out[t + "attn1.to_out.0.bias"] = (dim, dim), (dim,)
            out[t + "ff.net.0.proj.weight"], out[t + "ff.net.0.proj.bias"] = (8 * dim, dim), (8 * dim,)
            out[t + "ff.net.2.weight"], out[t + "ff.net.2.bias"] = (dim, 4 * dim), (dim,)
            out[t + "attn2.to_q.weight"] = (dim, dim)
            out[t + "attn2.to_k.weight"], out[t + "attn2.to_v.weight"] = (dim, ctx), (dim, ctx)
            out[t + "attn2.to_out.0.weight"], out[t + "attn2.to_out.0.bias"] = (dim, dim), (dim,)
            for nm in ("norm1", "norm2", "norm3"):
                out[t + nm + ".weight"], out[t + nm + ".bias"] = (dim,), (dim,)
            out[f + "proj_out.weight"], out[f + "proj_out.bias"] = (dim, dim, 1, 1), (dim,)
        out[p + "fn.norm.g"] = (1, dim, 1, 1)

    depth = len(ch_mult)
    mult = [1] + list(ch_mult)
    img_ctx = use_image_context and ctx > 0
    for i in range(depth):
        din, dout = nf * mult[i], nf * mult[i + 1]
        p = f"downs.{i}."
        resblock(p + "0.", din, din)
        resblock(p + "1.", din, din)
        attn(p + "2.", din, img_ctx and i >= 3)
        if i != depth - 1:
            out[p + "3.weight"], out[p + "3.bias"] = (dout, din, 4, 4), (dout,)
        else:
            out[p + "3.weight"] = (dout, din, 3, 3)
    for j in range(depth):                                # ups.insert(0, ...): ups[j] belongs to level depth-1-j
        i = depth - 1 - j
        din, dout = nf * mult[i], nf * mult[i + 1]
        p = f"ups.{j}."
        resblock(p + "0.", dout + din, dout)
        resblock(p + "1.", dout + din, dout)
        attn(p + "2.", dout, img_ctx and i >= 3)
        if i != 0:
            out[p + "3.1.weight"], out[p + "3.1.bias"] = (din, dout, 3, 3), (din,)
        else:
            out[p + "3.weight"] = (din, dout, 3, 3)
    mid = nf * mult[-1]
    resblock("mid_block1.", mid, mid)
    attn("mid_attn.", mid, img_ctx)
    resblock("mid_block2.", mid, mid)
    resblock("final_res_block.", 2 * nf, nf)
    out["final_conv.weight"], out["final_conv.bias"] = (out_nc, nf, 3, 3), (out_nc,)
    return out


def _key_seed(seed: int, key: str) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:6], "little")


def synth_state_dict(shapes: "OrderedDict[str, Tuple[int, ...]]", seed: int = 0,
                     dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random weights: every tensor gets its own generator keyed by name,
    so values do not depend on construction order.  Conv / linear weights are
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm gains 1+0.1*N(0,1); biases
    0.1*U(-1,1); the head is scaled by 0.05 so ``clamp(out+x,0,1)`` does not
    saturate ('saturating outputs')."""
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed(_key_seed(seed, key))
        is_norm = ".norm.weight" in key or key == "final_norm.weight" or key.endswith(".g") or \
            (".ln_" in key and key.endswith(".weight")) or re.search(r"norm\d\.weight$", key) is not None
        if is_norm:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float64)
        elif key.endswith("bias"):
            t = 0.1 * (2 * torch.rand(shape, generator=g, dtype=torch.float64) - 1)
        elif key.endswith(("class_embedding", "positional_embedding", ".proj")) and key.startswith("visual"):
            t = shape[-1 if key.endswith("embedding") else 0] ** -0.5 * torch.randn(shape, generator=g, dtype=torch.float64)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if key.startswith("upsamples."):
                fan_in = shape[0]  # ConvTranspose2d weight is (in, out, kh, kw)
            bound = 1.0 / (fan_in ** 0.5)
            t = bound * (2 * torch.rand(shape, generator=g, dtype=torch.float64) - 1)
            if key.startswith("head."):
                t = t * 0.05
        sd[key] = t.to(dtype)
    return sd


def synth_student_inputs(batch: int, height: int, width: int, cond_dim: int, num_masks: int,
                         mask_hw: Tuple[int, int] = (224, 224), seed: int = 1234):
    """SURVEY.md section 8(d) synthetic inputs: x ~ U[0,1), z_d ~ N(0,1), masks ~ U[0,1)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, height, width, generator=g)
    z = torch.randn(batch, cond_dim, generator=g)
    m = torch.rand(batch, num_masks, mask_hw[0], mask_hw[1], generator=g) if num_masks > 0 else None
    return x, z, m
