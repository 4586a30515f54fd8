"""``NAFNetWFIGate`` -- drop-in for the reference student net, forward on hand-written gfx950 kernels.

Mirrors /root/reference/underwater_ir/student/naf_unet_wfi/model.py:64-162: same constructor
signature, same ``forward(x, cond, masks=None) -> (restored, alpha_maps, deg_logits)``, same
``state_dict`` key names / shapes (the weight ABI, see synth.student_param_shapes), same
``ValueError`` when ``cond`` is None.  Extra, optional knob: ``precision`` ("fp32" exact-fp32
MFMA everywhere, "bf16" bf16 MFMA for GEMMs/attention with fp32 accumulation, norms, FFT).

Internally everything is NHWC (= torch channels_last memory) and runs through the C ABI in
``include/uwir.h``; this file only owns buffers, weight repacking and launch order.
Sizes: any even H and W, as in the reference (odd deeper levels shrink inside a stage's first block and the decoder
resizes to the skip, model.py:150-151; an odd H or W fails in the reference as well).  CUDA(HIP) tensors only.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .synth import student_param_shapes

_EPS = 1e-5
_LOG2E = 1.4426950408889634


class _Holder(nn.Module):
    """Parameter container; nesting reproduces the reference's dotted state_dict names."""


def _attach(root: nn.Module, dotted: str, shape) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Holder())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], nn.Parameter(torch.empty(shape)))


class NAFNetWFIGate(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, base_channels: int = 64, num_levels: int = 3,
                 blocks_per_level: int = 2, cond_dim: int = 16, num_masks: int = 0,
                 num_degradation_types: Optional[int] = None, attn_chunk_size: int = 256,
                 precision: str = "fp32") -> None:
        super().__init__()
        if base_channels % 4 != 0 or (base_channels // 4) % 4 != 0:
            raise ValueError("channels must be divisible by num_heads.") if base_channels % 4 else ValueError(
                "this build needs head_dim = base_channels/4 to be a multiple of 4")
        if base_channels * 2 ** (num_levels - 1) // 4 > 128:
            raise ValueError("head_dim > 128 is not supported by the attention kernel")
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.base_channels, self.num_levels, self.blocks_per_level = base_channels, num_levels, blocks_per_level
        self.cond_dim, self.num_masks = cond_dim, num_masks
        self.num_degradation_types = num_degradation_types
        self.attn_chunk_size = attn_chunk_size  # accepted for API parity; flash attention needs no chunking
        self.precision = precision
        self._shapes = student_param_shapes(in_channels, out_channels, base_channels, num_levels, blocks_per_level,
                                            cond_dim, num_masks, num_degradation_types)
        for key, shape in self._shapes.items():
            _attach(self, key, shape)
        self.reset_parameters()
        self._pack: Optional[dict] = None
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._streams: dict = {}            # device -> side stream of the SFFB branch
        self.overlap_branches = True        # False: every kernel of a forward on the caller's stream, in program order
        self.on_block = None                # optional callable(block_count), called after every WFI block is enqueued
        self.fold_norms = True              # bf16: LayerNorm2d affines folded into per-image weight panels (see _wfi_block)
        self.fuse_gate = True
        self.compose_reproj = True   # SFFB's last 1x1 conv composed with the CFC's k / vf convs (bf16 mode, no taps)
        self.chain_blocks = True     # blocks of one stage hand over (merged, hf) instead of idwt2 -> dwt2 (bf16 mode)               # bf16: WFIGate + idwt2 in one launch (ops.gate_idwt); False keeps the three launches
        self.fuse_ffn = True                # bf16, folded norms: ffn.0 -> GELU -> ffn.2 + residual in one launch (ops.ffn_fused)
        self.supports_windows = True        # forward(..., windows=...): halo windows read / written in place (tiling.py)

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self) -> None:
        for name, p in self.named_parameters():
            with torch.no_grad():
                if name.endswith("norm.weight") or name == "final_norm.weight":
                    p.fill_(1.0)
                elif name.endswith("norm.bias") or name == "final_norm.bias":
                    p.zero_()
                elif p.dim() > 1:
                    fan_in = p[0].numel() if not name.startswith("upsamples.") else p.shape[0] * 4
                    bound = 1.0 / math.sqrt(max(fan_in, 1))
                    p.uniform_(-bound, bound)
                else:
                    p.uniform_(-0.05, 0.05)

    def set_precision(self, precision: str) -> "NAFNetWFIGate":
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if precision != self.precision:
            self.precision = precision
            self._pack = None
        return self

    def replica(self) -> "NAFNetWFIGate":
        """A second set of activation buffers (and side stream) over THE SAME parameters and packed weights, so that two
        forwards can be in flight on two streams (pipeline.DualReplicaRestorer).  A snapshot: re-create it after
        load_state_dict / .to() / set_precision on the original."""
        import copy
        if self._pack is None:
            self._pack = self._build_pack()
        r = copy.copy(self)
        r._bufs, r._streams, r.on_block = {}, {}, None
        return r

    def _apply(self, fn, *a, **k):
        self._pack = None
        self._bufs = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._pack = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # ------------------------------------------------------------------ weight repacking (once)
    def _build_pack(self) -> dict:
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        dev = sd["stem.weight"].device
        wdt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        f32 = torch.float32

        def mat(w: torch.Tensor, dt=wdt) -> torch.Tensor:  # conv1x1 weight [N,K,1,1] -> [N,K]
            return w.reshape(w.shape[0], -1).to(dt).contiguous()

        def vec(b: torch.Tensor) -> torch.Tensor:
            return b.reshape(-1).to(f32).contiguous()

        def taps(w: torch.Tensor) -> torch.Tensor:  # depthwise [C,1,k,k] -> [k*k, C]
            return w.reshape(w.shape[0], -1).t().to(f32).contiguous()

        cond_w, cond_b = [], []
        cond_off = [0]

        def add_cond(w: torch.Tensor, b: torch.Tensor) -> Tuple[int, int]:
            off = cond_off[0]
            cond_w.append(w.to(f32))
            cond_b.append(b.to(f32))
            cond_off[0] += w.shape[0]
            return off, w.shape[0]

        M = self.num_masks
        Mp = ((M + 7) // 8) * 8 if M > 0 else 0
        pk: dict = {"Mp": Mp, "blocks": {}}

        def pack_block(p: str, Cc: int) -> dict:
            Cw = 3 * Cc
            b: dict = {"C": Cc}
            h = p + "hf_block."
            for nm in ("norm1", "norm2"):
                b[nm + "_g"] = vec(sd[h + nm + ".norm.weight"])
                b[nm + "_b"] = vec(sd[h + nm + ".norm.bias"])
                b[nm + "_ss"] = add_cond(sd[h + nm + ".linear.weight"], sd[h + nm + ".linear.bias"])
            b["qkv_w"], b["qkv_b"] = mat(sd[h + "qkv.weight"]), vec(sd[h + "qkv.bias"])
            if wdt == torch.bfloat16:    # fp32 masters of the two convs that follow a norm: per-image folded panels (forward)
                b["qkv_w32"], b["ffn0_w32"] = mat(sd[h + "qkv.weight"], f32), mat(sd[h + "ffn.0.weight"], f32)
            b["dw_t"] = tuple(taps(sd[h + f"dw_branches.{i}.weight"]) for i in range(3))
            b["dw_b"] = tuple(vec(sd[h + f"dw_branches.{i}.bias"]) for i in range(3))
            b["merge_w"], b["merge_b"] = mat(sd[h + "dw_merge.weight"]), vec(sd[h + "dw_merge.bias"])
            # q columns leave the GEMM already multiplied by (1/sqrt(hd))*log2(e) (attention works in exp2)
            cs = torch.ones(3 * Cw, device=dev, dtype=f32)
            cs[:Cw] = _LOG2E / math.sqrt(Cw // 12)
            b["merge_cs"] = cs
            b["proj_w"], b["proj_b"] = mat(sd[h + "proj.weight"]), vec(sd[h + "proj.bias"])
            b["ca_w0"], b["ca_b0"] = sd[h + "channel_attn.net.0.weight"].to(f32).contiguous(), vec(sd[h + "channel_attn.net.0.bias"])
            b["ca_w1"], b["ca_b1"] = sd[h + "channel_attn.net.2.weight"].to(f32).contiguous(), vec(sd[h + "channel_attn.net.2.bias"])
            b["ffn0_w"], b["ffn0_b"] = mat(sd[h + "ffn.0.weight"]), vec(sd[h + "ffn.0.bias"])
            b["ffn2_w"], b["ffn2_b"] = mat(sd[h + "ffn.2.weight"]), vec(sd[h + "ffn.2.bias"])
            if wdt == torch.bfloat16:
                b["ffn2_wp"] = ops.ffn_permute_w2(b["ffn2_w"])     # k order of the fused FFN kernel (csrc/ffn.hip)
            l = p + "lf_block."
            b["lnorm_g"], b["lnorm_b"] = vec(sd[l + "norm.norm.weight"]), vec(sd[l + "norm.norm.bias"])
            b["lnorm_ss"] = add_cond(sd[l + "norm.linear.weight"], sd[l + "norm.linear.bias"])
            b["ldw_t"] = tuple(taps(sd[l + f"spatial_convs.{i}.depthwise.weight"]) for i in range(3))
            b["ldw_b"] = tuple(vec(sd[l + f"spatial_convs.{i}.depthwise.bias"]) for i in range(3))
            # merge(cat_k pointwise_k(.)) is linear: fold pointwise into merge (fp64), sffb.py:46-47
            wm = sd[l + "merge.weight"].reshape(Cc, 3 * Cc).double()
            weff, beff = [], sd[l + "merge.bias"].double().clone()
            for i in range(3):
                wp = sd[l + f"spatial_convs.{i}.pointwise.weight"].reshape(Cc, Cc).double()
                bp = sd[l + f"spatial_convs.{i}.pointwise.bias"].double()
                wmk = wm[:, i * Cc:(i + 1) * Cc]
                weff.append(wmk @ wp)
                beff = beff + wmk @ bp
            b["lmerge_w"], b["lmerge_b"] = torch.cat(weff, dim=1).to(wdt).contiguous(), beff.to(f32).contiguous()
            # amplitude / phase 1x1 convs stay exact fp32: phase feeds cos/sin
            b["amp_w"], b["amp_b"] = mat(sd[l + "amp_mod.weight"], f32), vec(sd[l + "amp_mod.bias"])
            b["ph_w"], b["ph_b"] = mat(sd[l + "phase_mod.weight"], f32), vec(sd[l + "phase_mod.bias"])
            b["reproj_w"], b["reproj_b"] = mat(sd[l + "reproj.weight"]), vec(sd[l + "reproj.bias"])
            f = p + "cfc."
            wq, wvt = sd[f + "q_conv.weight"].reshape(Cc, Cc), sd[f + "vt_conv.weight"].reshape(Cc, Cc)
            # hf_sum = lh+hl+hh (model.py:37) folded into the weights: W @ (a+b+c) = [W|W|W] @ [a;b;c]
            b["qvt_w"] = torch.cat([wq.repeat(1, 3), wvt.repeat(1, 3)], dim=0).to(wdt).contiguous()
            b["qvt_b"] = torch.cat([vec(sd[f + "q_conv.bias"]), vec(sd[f + "vt_conv.bias"])]).contiguous()
            cs = torch.ones(2 * Cc, device=dev, dtype=f32)
            cs[:Cc] = _LOG2E / math.sqrt(Cc // 4)
            b["qvt_cs"] = cs
            b["kvf_w"] = torch.cat([sd[f + "k_conv.weight"].reshape(Cc, Cc), sd[f + "vf_conv.weight"].reshape(Cc, Cc)],
                                   dim=0).to(wdt).contiguous()
            b["kvf_b"] = torch.cat([vec(sd[f + "k_conv.bias"]), vec(sd[f + "vf_conv.bias"])]).contiguous()
            # k_conv / vf_conv act on lf = reproj(sf) with nothing in between (sffb.py:55, cfc.py:26-28): composed in fp64,
            # [k | vf] (reproj sf + b_r) + b = ([k | vf] reproj) sf + ([k | vf] b_r + b); used when lf itself is not asked for
            wkv = torch.cat([sd[f + "k_conv.weight"].reshape(Cc, Cc), sd[f + "vf_conv.weight"].reshape(Cc, Cc)], dim=0).double()
            b["kvfr_w"] = (wkv @ sd[l + "reproj.weight"].reshape(Cc, Cc).double()).to(wdt).contiguous()
            b["kvfr_b"] = (wkv @ sd[l + "reproj.bias"].double() + b["kvf_b"].double()).to(f32).contiguous()
            g = p + "gate."
            b["cp"] = add_cond(sd[g + "cond_proj.weight"], sd[g + "cond_proj.bias"])
            w0 = sd[g + "net.0.weight"].reshape(64, -1)  # columns [hf | lf | cond | masks], wfi_gate.py:37-42
            g1 = torch.zeros(64, 2 * Cc + Mp, device=dev, dtype=f32)
            g1[:, :2 * Cc] = w0[:, :2 * Cc]
            if M > 0:
                g1[:, 2 * Cc:2 * Cc + M] = w0[:, 3 * Cc:3 * Cc + M]
            b["g1_w"] = g1.to(wdt).contiguous()
            b["gc_w"], b["g1_b"] = w0[:, 2 * Cc:3 * Cc].to(f32).contiguous(), vec(sd[g + "net.0.bias"])
            b["g2_w"], b["g2_b"] = mat(sd[g + "net.2.weight"]), vec(sd[g + "net.2.bias"])
            return b

        ch = [self.base_channels * 2 ** i for i in range(self.num_levels)]
        for lvl in range(self.num_levels):
            for bi in range(self.blocks_per_level):
                p = f"encoder_stages.{lvl}.blocks.{bi}."
                pk["blocks"][p] = pack_block(p, ch[lvl])
        for di, lvl in enumerate(reversed(range(self.num_levels - 1))):
            for bi in range(self.blocks_per_level):
                p = f"decoder_stages.{di}.blocks.{bi}."
                pk["blocks"][p] = pack_block(p, ch[lvl])
        for lvl in range(self.num_levels - 1):
            w = sd[f"downsamples.{lvl}.weight"]  # [Cout, Cin, 2, 2] -> [Cout, (ky,kx,ci)]
            pk[f"down{lvl}_w"] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(wdt).contiguous()
            pk[f"down{lvl}_b"] = vec(sd[f"downsamples.{lvl}.bias"])
        for di in range(self.num_levels - 1):
            w = sd[f"upsamples.{di}.weight"]  # ConvTranspose2d [Cin, Cout, 2, 2] -> [(dy,dx,co), ci]
            pk[f"up{di}_w"] = w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).to(wdt).contiguous()
            pk[f"up{di}_b"] = vec(sd[f"upsamples.{di}.bias"]).repeat(4).contiguous()
            pk[f"fuse{di}_w"], pk[f"fuse{di}_b"] = mat(sd[f"skip_fusions.{di}.weight"]), vec(sd[f"skip_fusions.{di}.bias"])
        pk["stem_w"] = sd["stem.weight"].permute(2, 3, 1, 0).to(f32).contiguous()  # [3,3,Cin,Cout]
        pk["stem_b"] = vec(sd["stem.bias"])
        pk["head_w"] = sd["head.weight"].permute(0, 2, 3, 1).to(f32).contiguous()  # [Cout,3,3,Cin]
        pk["head_b"] = vec(sd["head.bias"])
        pk["fn_g"], pk["fn_b"] = vec(sd["final_norm.weight"]), vec(sd["final_norm.bias"])
        if self.num_degradation_types is not None:
            pk["dh_w0"], pk["dh_b0"] = mat(sd["degradation_head.net.0.weight"], f32), vec(sd["degradation_head.net.0.bias"])
            pk["dh_w1"], pk["dh_b1"] = mat(sd["degradation_head.net.2.weight"], f32), vec(sd["degradation_head.net.2.bias"])
        pk["cond_w"] = torch.cat(cond_w, dim=0).contiguous()
        pk["cond_b"] = torch.cat(cond_b, dim=0).contiguous()
        pk["adt"] = wdt
        return pk

    # ------------------------------------------------------------------ buffers
    def _buf(self, tag: str, rows: int, cols: int, dtype, dev, zero: bool = False) -> torch.Tensor:
        key = (tag, rows, cols, dtype)
        t = self._bufs.get(key)
        if t is None or t.device != dev:
            t = (torch.zeros if zero else torch.empty)(rows, cols, dtype=dtype, device=dev)
            self._bufs[key] = t
        return t

    def _side_stream(self, dev) -> "torch.cuda.Stream":
        st = self._streams.get(dev)
        if st is None:
            st = torch.cuda.Stream(device=dev)
            self._streams[dev] = st
        return st

    def _sffb(self, bp: dict, ll: torch.Tensor, B: int, h: int, w: int, cond_all: torch.Tensor, lvl: int,
              want_lf: bool = True):
        """SpatialFrequencyFusionBlock (sffb.py:44-55) on ll -> (lf, spatial).  Touches only buffers of its own (its
        LayerNorm workspace included), so it may run on another stream than the WTB branch.  ``want_lf=False`` returns
        the rows BEFORE the last 1x1 conv (reproj) in place of lf: the caller applies reproj composed with its own conv."""
        Cc = bp["C"]
        n, Mr = h * w, B * h * w
        dev, f32, adt = ll.device, torch.float32, self._pack["adt"]
        tg = f"L{lvl}"
        buf = lambda name, cols, dt=f32, zero=False: self._buf(tg + name, Mr, cols, dt, dev, zero)  # noqa: E731
        small = lambda name, cols: self._buf(tg + name, B, cols, f32, dev)  # noqa: E731
        ws = self._buf("lnws_sffb", 1, max(ops.ln2d_workspace_bytes(B, n, Cc) // 8, 1), torch.float64, dev)
        a3, s3 = small("a3", Cc), small("s3", Cc)
        o, ln_ = bp["lnorm_ss"]
        ops.ln2d_affine(ll, B, n, Cc, _EPS, bp["lnorm_g"], bp["lnorm_b"], cond_all[:, o:o + ln_], a3, s3, None, None, ws)
        ms2 = buf("ms2", 3 * Cc, adt)
        ops.dwconv_ms(ll, B, h, w, Cc, bp["ldw_t"], bp["ldw_b"], ms2, in_scale=a3, in_shift=s3)
        spatial = buf("spat", Cc)
        ops.conv_gemm(ms2, bp["lmerge_w"], spatial, B=B, H=h, W=w, C_=3 * Cc, N=Cc, bias=bp["lmerge_b"])
        amp, ph = buf("amp", Cc), buf("ph", Cc)
        fws = self._buf("fftws", 1, ops.fft_workspace_bytes(B, h, w, Cc) // 4, f32, dev)
        ops.fft2_amp_phase(spatial, B, h, w, Cc, amp, ph, fws)
        amp2, ph2 = buf("amp2", Cc), buf("ph2", Cc)
        ops.conv_gemm(amp, bp["amp_w"], amp2, B=B, H=h, W=w, C_=Cc, N=Cc, bias=bp["amp_b"])
        ops.conv_gemm(ph, bp["ph_w"], ph2, B=B, H=h, W=w, C_=Cc, N=Cc, bias=bp["ph_b"])
        sf = amp  # amp is dead: reuse
        ops.ifft2_polar_add(amp2, ph2, spatial, B, h, w, Cc, sf, fws)
        if not want_lf:
            return sf, spatial
        lf = buf("lf", Cc)
        ops.conv_gemm(sf, bp["reproj_w"], lf, B=B, H=h, W=w, C_=Cc, N=Cc, bias=bp["reproj_b"])
        return lf, spatial

    # ------------------------------------------------------------------ one WFI block
    def _wfi_block(self, bp: dict, x_in: Optional[torch.Tensor], B: int, h2: int, w2: int, cond_all: torch.Tensor,
                   gate_in: torch.Tensor, out: Optional[torch.Tensor], lvl: int, taps: Optional[dict]) -> torch.Tensor:
        """x_in: [B*h2*w2, ld] view with C channels; writes out ([B*2h*2w, ld'] view, h = h2 // 2, w = w2 // 2: an odd
        input side loses its last row / column in dwt2, dwt_fft.py:38); returns alpha [B,h,w,C].
        Chained blocks (bf16 mode, ``chain_blocks``): ``out=None`` leaves merged in the level's ll buffer and the block's hf
        rows where they are instead of forming idwt2's map, and ``x_in=None`` makes the next block start from those
        (dwt2 of idwt2 is the identity up to 1.2e-7, see uwir_gate_blend in include/uwir.h)."""
        Cc = bp["C"]
        Cw = 3 * Cc
        h, w = h2 // 2, w2 // 2
        n = h * w
        Mr = B * n
        dev, f32, adt = gate_in.device, torch.float32, self._pack["adt"]
        tg = f"L{lvl}"
        buf = lambda name, cols, dt=f32, zero=False: self._buf(tg + name, Mr, cols, dt, dev, zero)  # noqa: E731
        small = lambda name, cols: self._buf(tg + name, B, cols, f32, dev)  # noqa: E731
        # bf16 mode: the sub-band stream (ll, hf, x1, the block's hf output) is bf16 -- these rows are read by LayerNorm
        # statistics, the norm-and-cast pass, residual epilogues and the inverse transform, all HBM-bound; sums stay fp32
        ll, hf = buf("ll", Cc, adt), buf("hf", Cw, adt)
        if x_in is not None:
            ops.dwt2(x_in, B, h2, w2, Cc, ll, hf)
        ws = self._buf("lnws", 1, max(ops.ln2d_workspace_bytes(B, n, Cw) // 8, 1), torch.float64, dev)

        # The SFFB branch (on ll) and the WTB branch (on hf) are independent until the CFC: SFFB (FFT passes, C-wide 1x1
        # convs, a small depthwise conv) is enqueued on a second HIP stream next to WTB.  Measured 0.6 ms of the 57.7 ms
        # step, bit-identical.  (Releasing it exactly at the attention launch, to hide it in the shadow of that
        # exp-bound kernel, gains the same 0.6 ms but slows the attention launch itself by 2 %: its blocks fill the
        # register file, so foreign workgroups only get in as its rounds retire.)
        # bf16 mode without taps: lf = reproj(sf) is only ever read by the CFC's k / vf convs, so the two 1x1 convs run as one
        # (weights composed at load time) and the fp32 lf rows are never written
        want_lf = not (adt == torch.bfloat16 and taps is None and self.compose_reproj)
        side = self._side_stream(dev) if self.overlap_branches else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))   # ll / hf of THIS block are written
            with torch.cuda.stream(side):
                lf, spatial = self._sffb(bp, ll, B, h, w, cond_all, lvl, want_lf)
                sffb_done = torch.cuda.Event()
                sffb_done.record(side)

        # ---------------- WideTransformerBlock (wtb.py:83-124) on hf
        a1, s1, pooled = small("a1", Cw), small("s1", Cw), small("pool", Cw)
        o, ln_ = bp["norm1_ss"]
        ops.ln2d_affine(hf, B, n, Cw, _EPS, bp["norm1_g"], bp["norm1_b"], cond_all[:, o:o + ln_], a1, s1, None, pooled, ws)
        hid_ca = small("cah", bp["ca_w0"].shape[0])
        cw = small("caw", Cw)
        ops.linear(pooled, bp["ca_w0"], bp["ca_b0"], hid_ca, ops.ACT_RELU)
        ops.linear(hid_ca, bp["ca_w1"], bp["ca_b1"], cw, ops.ACT_SIGMOID)
        rs, rh = small("rs", Cw), small("rh", Cw)
        ops.chan_attn_fold(a1, s1, cw, rs, rh)
        qkv_base = buf("qkvb", 3 * Cw, adt)
        bf = adt == torch.bfloat16
        # bf16: the norm's per-image affine is folded into per-image weight panels W diag(a_b) and bias rows W s_b + bias
        # (ops.fold_affine_weights, a few microseconds), so the LDS-DMA GEMM reads the un-normalised bf16 rows directly; the
        # norm-and-cast pass (affine_cast: a full round trip of the activation) is only the fallback for shapes the
        # batched-weight GEMM does not take.  fp32: normalise-on-load in the register-staged GEMM.
        fold = bf and self.fold_norms and n % 128 == 0 and Cw % 64 == 0
        if fold:
            wb = self._buf(tg + "wb_qkv", B * 3 * Cw, Cw, adt, dev).view(B, 3 * Cw, Cw)
            bbq = small("bb_qkv", 3 * Cw)
            ops.fold_affine_weights(bp["qkv_w32"], a1, s1, bp["qkv_b"], wb, bbq)
            ops.conv_gemm(hf, wb, qkv_base, B=B, H=h, W=w, C_=Cw, N=3 * Cw, batch_bias=bbq, w_batched=True)
        elif bf:
            xn = buf("xn", Cw, adt)
            ops.affine_cast(hf, B, n, Cw, a1, s1, xn)
            ops.conv_gemm(xn, bp["qkv_w"], qkv_base, B=B, H=h, W=w, C_=Cw, N=3 * Cw, bias=bp["qkv_b"])
        else:
            ops.conv_gemm(hf, bp["qkv_w"], qkv_base, B=B, H=h, W=w, C_=Cw, N=3 * Cw, a_scale=a1, a_shift=s1,
                          bias=bp["qkv_b"])
        ms = buf("ms", 9 * Cw, adt)
        ops.dwconv_ms(qkv_base, B, h, w, 3 * Cw, bp["dw_t"], bp["dw_b"], ms)
        qkv = buf("qkv", 3 * Cw, adt)
        ops.conv_gemm(ms, bp["merge_w"], qkv, B=B, H=h, W=w, C_=9 * Cw, N=3 * Cw, bias=bp["merge_b"],
                      col_scale=bp["merge_cs"])
        att = buf("att", Cw, adt)
        hd = Cw // 12
        ops.flash_attn(qkv[:, :Cw], qkv[:, Cw:2 * Cw], qkv[:, 2 * Cw:], None, B, n, 12, hd, 0.0, att, None)
        x1 = buf("x1", Cw, adt)
        ops.conv_gemm(att, bp["proj_w"], x1, B=B, H=h, W=w, C_=Cw, N=Cw, bias=bp["proj_b"], res=hf, res_scale=rs, res_shift=rh)
        a2, s2 = small("a2", Cw), small("s2", Cw)
        o, ln_ = bp["norm2_ss"]
        ops.ln2d_affine(x1, B, n, Cw, _EPS, bp["norm2_g"], bp["norm2_b"], cond_all[:, o:o + ln_], a2, s2, None, None, ws)
        hf_out = hf  # hf is dead after the proj GEMM: reuse its storage
        fused_ffn = fold and self.fuse_ffn and ops.ffn_fused_supported(Cw, n)
        hidden = None if fused_ffn else buf("ffh", 2 * Cw, adt)
        if fold:
            wb = self._buf(tg + "wb_ffn", B * 2 * Cw, Cw, adt, dev).view(B, 2 * Cw, Cw)
            bbf = small("bb_ffn", 2 * Cw)
            ops.fold_affine_weights(bp["ffn0_w32"], a2, s2, bp["ffn0_b"], wb, bbf)
            if fused_ffn:   # ffn.0 -> GELU -> ffn.2 + residual in one launch: the [M, 2Cw] hidden rows never reach HBM
                ops.ffn_fused(x1, wb, bbf, bp["ffn2_wp"], bp["ffn2_b"], B, n, Cw, hf_out)
            else:
                ops.conv_gemm(x1, wb, hidden, B=B, H=h, W=w, C_=Cw, N=2 * Cw, batch_bias=bbf, act=ops.ACT_GELU, w_batched=True)
        elif bf:
            ops.affine_cast(x1, B, n, Cw, a2, s2, xn)
            ops.conv_gemm(xn, bp["ffn0_w"], hidden, B=B, H=h, W=w, C_=Cw, N=2 * Cw, bias=bp["ffn0_b"], act=ops.ACT_GELU)
        else:
            ops.conv_gemm(x1, bp["ffn0_w"], hidden, B=B, H=h, W=w, C_=Cw, N=2 * Cw, a_scale=a2, a_shift=s2,
                          bias=bp["ffn0_b"], act=ops.ACT_GELU)
        if not fused_ffn:
            ops.conv_gemm(hidden, bp["ffn2_w"], hf_out, B=B, H=h, W=w, C_=2 * Cw, N=Cw, bias=bp["ffn2_b"], res=x1)

        if side is not None:
            torch.cuda.current_stream(dev).wait_event(sffb_done)
        else:
            lf, spatial = self._sffb(bp, ll, B, h, w, cond_all, lvl, want_lf)

        # ---------------- CrossFrequencyConditioner (cfc.py:24-39)
        qvt, kvf = buf("qvt", 2 * Cc, adt), buf("kvf", 2 * Cc, adt)
        ops.conv_gemm(hf_out, bp["qvt_w"], qvt, B=B, H=h, W=w, C_=Cw, N=2 * Cc, bias=bp["qvt_b"], col_scale=bp["qvt_cs"])
        if want_lf:
            ops.conv_gemm(lf, bp["kvf_w"], kvf, B=B, H=h, W=w, C_=Cc, N=2 * Cc, bias=bp["kvf_b"])
        else:
            ops.conv_gemm(lf, bp["kvfr_w"], kvf, B=B, H=h, W=w, C_=Cc, N=2 * Cc, bias=bp["kvfr_b"])
        hd2 = Cc // 4
        ops.flash_attn(qvt[:, :Cc], kvf[:, :Cc], qvt[:, Cc:], kvf[:, Cc:], B, n, 4, hd2, 0.0,
                       gate_in[:, :Cc], gate_in[:, Cc:2 * Cc])

        # ---------------- WFIGate (wfi_gate.py:29-45)
        o, ln_ = bp["cp"]
        bb = small("gbb", 64)
        ops.linear(cond_all[:, o:o + ln_], bp["gc_w"], bp["g1_b"], bb)
        Kg = gate_in.shape[1]
        alpha = torch.empty(B, h, w, Cc, dtype=f32, device=dev)
        # bf16: gate MLP, blend and the inverse transform in one launch (csrc/gate.hip): the hidden rows, the merged rows
        # and the second read of t / f never reach HBM (3.0 -> 1.7 KB per sub-band pixel at C = 64)
        if out is None:     # chained: the next block of this stage reads merged (-> ll) and hf_out (= hf) in place
            ops.gate_blend(gate_in, bp["g1_w"], bp["g2_w"], bb, bp["g2_b"], B, h, w, Cc, alpha, ll)
            return alpha
        fused_gate = bf and self.fuse_gate and out.dtype == torch.bfloat16 and ops.gate_idwt_supported(Cc, Kg)
        if fused_gate:
            merged = buf("merged", Cc) if taps is not None else None
            ops.gate_idwt(gate_in, bp["g1_w"], bp["g2_w"], bb, bp["g2_b"], hf_out, B, h, w, Cc, out, alpha, merged)
        else:
            ghid = buf("ghid", 64, adt)
            ops.conv_gemm(gate_in, bp["g1_w"], ghid, B=B, H=h, W=w, C_=Kg, N=64, batch_bias=bb, act=ops.ACT_RELU)
            merged = buf("merged", Cc)
            ops.conv_gemm(ghid, bp["g2_w"], alpha.view(Mr, Cc), B=B, H=h, W=w, C_=64, N=Cc, bias=bp["g2_b"],
                          act=ops.ACT_SIGMOID, blend_t=gate_in[:, :Cc], blend_f=gate_in[:, Cc:2 * Cc], blend_out=merged)
            ops.idwt2(merged, hf_out, B, h, w, Cc, out)
        if taps is not None:
            nchw = lambda t, c: t[:, :c].reshape(B, h, w, c).permute(0, 3, 1, 2).float().clone()  # noqa: E731
            taps.update(hf=nchw(hf_out, Cw), lf=nchw(lf, Cc), t_out=nchw(gate_in, Cc),
                        f_out=nchw(gate_in[:, Cc:], Cc), merged=nchw(merged, Cc), alpha=alpha.permute(0, 3, 1, 2),
                        out=out[:, :Cc].reshape(B, 2 * h, 2 * w, Cc).permute(0, 3, 1, 2).float().clone(),
                        wtb=dict(qkv_base=nchw(qkv_base, 3 * Cw), att=nchw(att, Cw),
                                 x1=nchw(x1, Cw)),
                        sffb=dict(spatial=nchw(spatial, Cc), freq=None))
        return alpha

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    @ops.on_operand_device
    def forward(self, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor] = None, taps: Optional[dict] = None,
                windows: Optional[dict] = None) -> Tuple[torch.Tensor, List[torch.Tensor], Optional[torch.Tensor]]:
        """``windows`` (halo tiling, tiling.TiledRestorer; no reference counterpart): dict(win=int32 [nwin, 8] device
        descriptors (uwir_window), T=window edge, out=output frames).  Then ``x`` / ``masks`` are whole NCHW FRAMES, the
        batch the network sees is the nwin windows read in place by the stem / mask-resize kernels, ``cond`` has one row
        per window, and the head writes every window's owned core straight into ``windows["out"]`` (returned)."""
        if cond is None:
            raise ValueError("Condition vector z_d must be provided.")
        if not x.is_cuda:
            raise RuntimeError("NAFNetWFIGate (MI355X build) runs on a ROCm device only; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError("Input must be (B, in_channels, H, W).")
        B, _, H, W = x.shape
        win = None
        if windows is not None:
            win, T = windows["win"], int(windows["T"])
            if masks is not None and tuple(masks.shape[-2:]) != (H, W):
                raise ValueError("window mode: masks must be frame sized (B, M, H, W)")
            if windows["out"].shape != x.shape or taps is not None:
                raise ValueError("window mode: out must have the frames' shape; taps are not available")
            B, H, W = int(win.shape[0]), T, T
            if cond.shape[0] != B:
                raise ValueError("window mode: cond needs one row per window")
        # Level geometry.  dwt2 is a stride-2 valid conv (dwt_fft.py:38): the first block of a stage turns an odd input side
        # into the even side below it, everything after it at that level (later blocks, the skip, the down-sampling conv) has
        # that even size, and the decoder resizes its up-sampled map to the skip's size when they differ (model.py:150-151).
        # Odd H or W itself fails in the reference too: the head's output is one pixel short of the residual (model.py:161).
        if H % 2 or W % 2:
            raise ValueError(f"H and W must be even (got {H}x{W}): with an odd side the reference's final "
                             "`out + residual` (model.py:161) adds tensors of different sizes")
        size_in, size = [], []
        hi, wi = H, W
        for lvl in range(self.num_levels):
            if hi < 2 or wi < 2:
                raise ValueError(f"a {H}x{W} image is too small for {self.num_levels} levels")
            size_in.append((hi, wi))
            size.append((hi // 2 * 2, wi // 2 * 2))
            hi, wi = size[-1][0] // 2, size[-1][1] // 2
        if (masks is None) != (self.num_masks == 0):
            raise ValueError("masks must be given iff the model was built with num_masks > 0")
        if masks is not None and masks.shape[1] != self.num_masks:
            raise ValueError("masks channel count differs from num_masks")
        if self._pack is None:
            self._pack = self._build_pack()
        pk = self._pack
        dev, f32 = x.device, torch.float32
        x = x.contiguous().to(f32)
        cond = cond.contiguous().to(f32)
        masks = None if masks is None else masks.contiguous().to(f32)
        L, ch = self.num_levels, [self.base_channels * 2 ** i for i in range(self.num_levels)]
        Mp, Mk = pk["Mp"], self.num_masks

        cond_all = self._buf("cond_all", B, pk["cond_w"].shape[0], f32, dev)
        ops.linear(cond, pk["cond_w"], pk["cond_b"], cond_all)

        # per level: gate input [t | f | masks(resized) | 0-pad], feature / concat buffers
        gate_in, feat, cat = [], [], []
        for lvl in range(L):
            (Hi, Wi), (Hl, Wl) = size_in[lvl], size[lvl]
            h, w = Hl // 2, Wl // 2
            gi = self._buf(f"gate_in{lvl}", B * h * w, 2 * ch[lvl] + Mp, f32, dev, zero=True)
            if masks is not None and win is not None:
                ops.bilinear_win(masks, win, H, gi[:, 2 * ch[lvl]:2 * ch[lvl] + Mk], h, w)
            elif masks is not None:
                ops.bilinear_nchw_to_nhwc(masks, gi[:, 2 * ch[lvl]:2 * ch[lvl] + Mk], h, w)
            gate_in.append(gi)
            # (bf16 mode: the feature stream between blocks, the skip concats and the sub-bands are bf16 rows)
            # (feat rows: the stage's input size; everything a block writes has the even size and uses the leading rows)
            feat.append([self._buf(f"feat{lvl}_{i}", B * Hi * Wi, ch[lvl], pk["adt"], dev) for i in range(2)])
            cat.append(self._buf(f"cat{lvl}", B * Hl * Wl, 2 * ch[lvl], pk["adt"], dev) if lvl < L - 1 else None)

        if win is not None:
            ops.stem_conv3x3_win(x, win, H, pk["stem_w"], pk["stem_b"], feat[0][0])
        else:
            ops.stem_conv3x3(x, pk["stem_w"], pk["stem_b"], feat[0][0])
        if taps is not None:
            taps["stem"] = feat[0][0].reshape(B, H, W, ch[0]).permute(0, 3, 1, 2).float().clone()
        alphas: List[torch.Tensor] = []

        def run_stage(prefix: str, lvl: int, cur: torch.Tensor, final_out: Optional[torch.Tensor],
                      in_size: Tuple[int, int]) -> torch.Tensor:
            Hl, Wl = size[lvl]
            chained = False
            for bi in range(self.blocks_per_level):
                last = bi == self.blocks_per_level - 1
                if last and final_out is not None:
                    dst = final_out
                else:
                    dst = (feat[lvl][1] if cur.data_ptr() == feat[lvl][0].data_ptr() else feat[lvl][0])[:B * Hl * Wl]
                bt = {} if taps is not None else None
                hb, wb_ = in_size if bi == 0 else (Hl, Wl)     # only a stage's first block can see an odd side
                # bf16 mode: a block followed by another block of the stage hands over (merged, hf) instead of idwt2's map
                chain_out = (not last and taps is None and self.chain_blocks and self.fuse_gate
                             and pk["adt"] == torch.bfloat16
                             and ops.gate_idwt_supported(ch[lvl], gate_in[lvl].shape[1]))
                a = self._wfi_block(pk["blocks"][f"{prefix}.blocks.{bi}."], None if chained else cur, B, hb, wb_, cond_all,
                                    gate_in[lvl], None if chain_out else dst, lvl, bt)
                chained = chain_out
                alphas.append(a.permute(0, 3, 1, 2))
                if self.on_block is not None:
                    self.on_block(len(alphas))     # pipeline.DualReplicaRestorer staggers its second stream on this
                if taps is not None:
                    taps[f"{prefix}.blocks.{bi}"] = bt
                cur = dst
            return cur

        cur = feat[0][0]
        for lvl in range(L):
            Hl, Wl = size[lvl]
            skip_view = cat[lvl][:, ch[lvl]:] if lvl < L - 1 else None  # encoder output lands in [x_up | skip]
            cur = run_stage(f"encoder_stages.{lvl}", lvl, cur, skip_view, size_in[lvl])
            if lvl < L - 1:
                nxt = feat[lvl + 1][0]
                ops.conv_gemm(cur, pk[f"down{lvl}_w"], nxt, B=B, H=Hl, W=Wl, C_=ch[lvl], N=ch[lvl + 1], KH=2, KW=2,
                              stride=2, pad=0, bias=pk[f"down{lvl}_b"])
                cur = nxt
        logits = None
        if self.num_degradation_types is not None:
            Hb, Wb = size[L - 1]
            gap = self._buf("gap", B, ch[-1], f32, dev)
            ws = self._buf("lnws", 1, max(ops.ln2d_workspace_bytes(B, Hb * Wb, ch[-1]) // 8, 1), torch.float64, dev)
            ops.ln2d_affine(cur, B, Hb * Wb, ch[-1], _EPS, None, None, None, None, None, gap, None, ws)
            dh = self._buf("dh_hid", B, 256, f32, dev)
            ops.linear(gap, pk["dh_w0"], pk["dh_b0"], dh, ops.ACT_GELU)
            logits = torch.empty(B, self.num_degradation_types, dtype=f32, device=dev)
            ops.linear(dh, pk["dh_w1"], pk["dh_b1"], logits)
        for di, lvl in enumerate(reversed(range(L - 1))):
            (Hs, Ws), (Hl, Wl) = size[lvl + 1], size[lvl]
            # ConvTranspose2d(k=2,s=2) = GEMM + depth-to-space scatter into the left half of the concat buffer -- or, when
            # the skip is larger (an odd side below it), into a buffer of its own that is then resized (model.py:150-151)
            direct = (2 * Hs, 2 * Ws) == (Hl, Wl)
            up_dst = cat[lvl][:, :ch[lvl]] if direct else self._buf(f"up_tmp{lvl}", B * 4 * Hs * Ws, ch[lvl], pk["adt"], dev)
            ops.conv_gemm(cur, pk[f"up{di}_w"], up_dst, B=B, H=Hs, W=Ws, C_=ch[lvl + 1], N=4 * ch[lvl],
                          bias=pk[f"up{di}_b"], store_mode=1)
            if not direct:
                ops.bilinear_rows(up_dst, B, 2 * Hs, 2 * Ws, ch[lvl], cat[lvl][:, :ch[lvl]], Hl, Wl)
            fused = feat[lvl][0][:B * Hl * Wl]
            ops.conv_gemm(cat[lvl], pk[f"fuse{di}_w"], fused, B=B, H=Hl, W=Wl, C_=2 * ch[lvl], N=ch[lvl],
                          bias=pk[f"fuse{di}_b"])
            cur = run_stage(f"decoder_stages.{di}", lvl, fused, None, (Hl, Wl))
        fa, fs = self._buf("fn_a", B, ch[0], f32, dev), self._buf("fn_s", B, ch[0], f32, dev)
        ws = self._buf("lnws", 1, max(ops.ln2d_workspace_bytes(B, H * W, ch[0]) // 8, 1), torch.float64, dev)
        ops.ln2d_affine(cur, B, H * W, ch[0], _EPS, pk["fn_g"], pk["fn_b"], None, fa, fs, None, None, ws)
        if self.in_channels != self.out_channels:
            raise ValueError("in_channels must equal out_channels (the head adds the input residual, model.py:161)")
        if win is not None:
            out = windows["out"]
            ops.head_conv3x3_win(cur, H, ch[0], fa, fs, pk["head_w"], pk["head_b"], self.out_channels, x, out, win)
            return out, alphas, logits
        out = torch.empty(B, self.out_channels, H, W, dtype=f32, device=dev)
        pre = torch.empty_like(out) if taps is not None else None
        ops.head_conv3x3(cur, B, H, W, ch[0], fa, fs, pk["head_w"], pk["head_b"], self.out_channels, x, out, pre)
        if taps is not None:
            taps["pre_clamp"] = pre
        return out, alphas, logits
