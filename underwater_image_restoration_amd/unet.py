"""``ConditionalUNet`` -- drop-in for the IR-SDE score network, forward on hand-written gfx950 kernels.

Mirrors /root/reference/legacy/third_party/universal-image-restoration/config/daclip-sde/models/modules/
DenoisingUNet_arch.py:21-174: same constructor, same ``forward(xt, cond, time, text_context=None,
image_context=None)`` (NCHW in, NCHW noise prediction out), same ``state_dict`` keys / shapes
(synth.unet_param_shapes; a ``module.`` prefix from nn.DataParallel checkpoints is stripped like
models/base_model.py:98-103).  Extra knob: ``precision`` ("fp32": exact-fp32 MFMA, fp32 activations; "bf16": bf16 MFMA
with fp32 accumulation, and the activation stream between kernels -- ResBlock / attention outputs, skips, residuals --
is stored as bf16 too: at level 0 the 3x3 convs are HBM-bound on exactly those tensors.  Norm statistics, softmaxes and
every accumulation stay fp32.  Measured against the fp32 model (tools/exp/bf16_stream_emul.py, B = 2, 256 x 256): one
step rel. L2 5.0e-3 with an fp32 stream, 5.9e-3 with the bf16 stream; 100-step reverse_posterior 49.3 vs 49.0 dB).

Everything runs NHWC through the C ABI (include/uwir.h); this file owns buffers, weight repacking, launch order:
  * 3x3 / 4x4-s2 / 7x7 / nearest-2x+3x3 convs = implicit GEMM on MFMA; the ResBlock's ``x*(scale+1)+shift`` + SiLU and
    the ``+ res_conv(x)`` live in the GEMM epilogue; all ResBlock time-MLPs are ONE small linear per forward.
  * skip concats are never materialised: producers write straight into column slices of the concat buffers.
  * LinearAttention = 3 HBM passes (uwir_linear_attention); SpatialTransformer = GroupNorm folded into the
    proj_in GEMM, flash attention, cross-attention on the single context token collapsed to a per-sample bias
    (softmax over one key is 1: attention.py:176-190 with context (B,1,512)), GEGLU.
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .student import _Holder, _attach  # noqa: F401
from .synth import unet_param_shapes

_LOG2E = 1.4426950408889634


class ConditionalUNet(nn.Module):
    def __init__(self, in_nc, out_nc, nf, ch_mult=[1, 2, 4, 4], context_dim=512, use_degra_context=True,
                 use_image_context=False, upscale=1, precision: str = "fp32") -> None:
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if nf % 32:
            raise ValueError("nf must be a multiple of 32 (the reference derives head counts as dim // 32)")
        if 2 * in_nc > 8 or out_nc > 4:
            raise ValueError("this build supports in_nc <= 4 and out_nc <= 4")
        self.in_nc, self.out_nc, self.nf, self.ch_mult = in_nc, out_nc, nf, list(ch_mult)
        self.depth = len(ch_mult)
        self.upscale = upscale  # not used (as in the reference)
        self.context_dim = -1 if context_dim is None else context_dim
        self.use_degra_context, self.use_image_context = use_degra_context, use_image_context
        self.precision = precision
        self._has_prompt = self.context_dim > 0 and use_degra_context
        self._has_st = self.context_dim > 0 and use_image_context
        shapes = unet_param_shapes(in_nc, out_nc, nf, ch_mult, context_dim, use_degra_context, use_image_context)
        for key, shape in shapes.items():
            _attach(self, key, shape)
        with torch.no_grad():
            for name, p in self.named_parameters():
                parts = name.split(".")
                if name.endswith(".g") or (len(parts) > 1 and "norm" in parts[-2] and parts[-1] == "weight"):
                    p.fill_(1.0)
                elif name.endswith("bias") or "proj_out" in name:
                    p.zero_()      # zero_module(proj_out), attention.py:244-248
                elif name == "prompt":
                    p.uniform_(0, 1)
                else:
                    fan_in = p[0].numel()
                    p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        self._pack: Optional[dict] = None
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self.precise_tail = True             # bf16: catF / final res_conv / final ResBlock output in fp32 (see forward)
        self.fused_linattn = True            # bf16: Residual(PreNorm(LinearAttention)) as two fused passes (csrc/linattn.hip)
        self._emb_cache: Dict[tuple, torch.Tensor] = {}
        self._schedule: Optional[dict] = None   # conditioning_schedule()

    # ------------------------------------------------------------------ parameters
    def set_precision(self, precision: str) -> "ConditionalUNet":
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if precision != self.precision:
            self.precision, self._pack = precision, None
        return self

    def _apply(self, fn, *a, **k):
        self._pack, self._bufs, self._emb_cache = None, {}, {}
        self._schedule = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._pack = None
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict, **kw)

    def _levels(self):
        mult = [1] + self.ch_mult
        return [(self.nf * mult[i], self.nf * mult[i + 1]) for i in range(self.depth)]

    def _build_pack(self) -> dict:
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        wdt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        f32 = torch.float32
        dev = sd["init_conv.weight"].device
        td = 4 * self.nf

        def conv(w: torch.Tensor, cpad: int = 0) -> torch.Tensor:  # [N,C,kh,kw] -> [N, kh*kw*C]
            if cpad > w.shape[1]:
                w = torch.cat([w, w.new_zeros(w.shape[0], cpad - w.shape[1], w.shape[2], w.shape[3])], dim=1)
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(wdt).contiguous()

        def vec(b: torch.Tensor) -> torch.Tensor:
            return b.reshape(-1).to(f32).contiguous()

        pk: dict = {"adt": wdt}
        ss_w, ss_b, ss_off = [], [], [0]

        def resblock(p: str) -> dict:
            w1 = sd[p + "block1.proj.weight"]
            dout = w1.shape[0]
            bias = sd[p + "mlp.1.bias"].to(f32).clone()
            bias[:dout] += 1.0                       # x*(scale+1)+shift: the "+1" rides in the linear's bias
            ss_w.append(sd[p + "mlp.1.weight"].to(f32))
            ss_b.append(bias)
            off = ss_off[0]
            ss_off[0] += 2 * dout
            r = dict(din=w1.shape[1], dout=dout, off=off, w1=conv(w1), w2=conv(sd[p + "block2.proj.weight"]))
            r["wres"] = conv(sd[p + "res_conv.weight"]) if (p + "res_conv.weight") in sd else None
            return r

        def conv32(w: torch.Tensor) -> torch.Tensor:
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(f32).contiguous()

        ov_w, ov_b, ov_off = [], [], [0]

        def geglu_rows(w: torch.Tensor) -> torch.Tensor:
            # [value rows | gate rows] -> groups of 16 = 8 value rows then their 8 gate rows (conv_gemm store_mode 2)
            F_ = w.shape[0] // 2
            return torch.stack([w[:F_].reshape(F_ // 8, 8, *w.shape[1:]), w[F_:].reshape(F_ // 8, 8, *w.shape[1:])],
                               dim=1).reshape(w.shape)

        def attn(p: str) -> dict:
            a: dict = {"g_pre": vec(sd[p + "fn.norm.g"])}
            f = p + "fn.fn."
            if (f + "to_qkv.weight") in sd:
                a.update(kind="linear", w_qkv=conv(sd[f + "to_qkv.weight"]), w_out=conv(sd[f + "to_out.0.weight"]),
                         b_out=vec(sd[f + "to_out.0.bias"]), g_out=vec(sd[f + "to_out.1.g"]))
                return a
            t = f + "transformer_blocks.0."
            dim = sd[f + "proj_in.weight"].shape[0]
            cs = torch.ones(3 * dim, device=dev, dtype=f32)
            cs[:dim] = _LOG2E / math.sqrt(32)
            # cross-attention on ONE context token: softmax == 1, so attn2(x, ctx) = to_out(to_v(ctx)) for every query
            wov = sd[t + "attn2.to_out.0.weight"].double() @ sd[t + "attn2.to_v.weight"].double()
            ov_w.append(wov.to(f32))
            ov_b.append(sd[t + "attn2.to_out.0.bias"].to(f32))
            off = ov_off[0]
            ov_off[0] += dim
            a.update(kind="spatial", dim=dim, heads=dim // 32, ov_off=off,
                     gn_w=vec(sd[f + "norm.weight"]), gn_b=vec(sd[f + "norm.bias"]),
                     w_in=conv(sd[f + "proj_in.weight"]), b_in=vec(sd[f + "proj_in.bias"]),
                     w_qkv=torch.cat([sd[t + "attn1.to_q.weight"], sd[t + "attn1.to_k.weight"],
                                      sd[t + "attn1.to_v.weight"]], dim=0).to(wdt).contiguous(), qcs=cs,
                     w_o1=sd[t + "attn1.to_out.0.weight"].to(wdt).contiguous(), b_o1=vec(sd[t + "attn1.to_out.0.bias"]),
                     n1=(vec(sd[t + "norm1.weight"]), vec(sd[t + "norm1.bias"])),
                     n3=(vec(sd[t + "norm3.weight"]), vec(sd[t + "norm3.bias"])),
                     w_ff0=geglu_rows(sd[t + "ff.net.0.proj.weight"]).to(wdt).contiguous(),
                     b_ff0=geglu_rows(sd[t + "ff.net.0.proj.bias"]).to(f32).contiguous(),
                     w_ff2=sd[t + "ff.net.2.weight"].to(wdt).contiguous(), b_ff2=vec(sd[t + "ff.net.2.bias"]),
                     w_pout=conv(sd[f + "proj_out.weight"]), b_pout=vec(sd[f + "proj_out.bias"]))
            return a

        pk["init_w"] = conv(sd["init_conv.weight"], cpad=8)
        pk["downs"], pk["ups"] = [], []
        for i in range(self.depth):
            p = f"downs.{i}."
            d = dict(b1=resblock(p + "0."), b2=resblock(p + "1."), attn=attn(p + "2."), w_down=conv(sd[p + "3.weight"]),
                     b_down=vec(sd[p + "3.bias"]) if (p + "3.bias") in sd else None)
            pk["downs"].append(d)
        pk["mid1"], pk["mid_attn"], pk["mid2"] = resblock("mid_block1."), attn("mid_attn."), resblock("mid_block2.")
        for j in range(self.depth):
            i = self.depth - 1 - j
            p = f"ups.{j}."
            key = p + ("3.1." if i != 0 else "3.")
            u = dict(b1=resblock(p + "0."), b2=resblock(p + "1."), attn=attn(p + "2."), w_up=conv(sd[key + "weight"]),
                     b_up=vec(sd[key + "bias"]) if (key + "bias") in sd else None)
            pk["ups"].append(u)
        pk["final"] = resblock("final_res_block.")
        if "final_res_block.res_conv.weight" in sd:     # precise tail path (bf16 mode): the last 1x1 at fp32
            pk["final"]["wres32"] = conv32(sd["final_res_block.res_conv.weight"])
        pk["tail_w"] = sd["final_conv.weight"].permute(0, 2, 3, 1).to(f32).contiguous()  # [Cout,3,3,Cin]
        pk["tail_b"] = vec(sd["final_conv.bias"])
        pk["ss_w"], pk["ss_b"] = torch.cat(ss_w, dim=0).contiguous(), torch.cat(ss_b, dim=0).contiguous()
        pk["t1_w"], pk["t1_b"] = sd["time_mlp.1.weight"].to(f32).contiguous(), vec(sd["time_mlp.1.bias"])
        pk["t3_w"], pk["t3_b"] = sd["time_mlp.3.weight"].to(f32).contiguous(), vec(sd["time_mlp.3.bias"])
        if self._has_prompt:
            pk["x0_w"], pk["x0_b"] = sd["text_mlp.0.weight"].to(f32).contiguous(), vec(sd["text_mlp.0.bias"])
            pk["x2_w"], pk["x2_b"] = sd["text_mlp.2.weight"].to(f32).contiguous(), vec(sd["text_mlp.2.bias"])
            pk["prompt"] = vec(sd["prompt"])
            # t = time_mlp.3(g) + prompt_mlp(pe)  ==  [W3 | Wpm] @ [g ; pe] + (b3 + bpm)
            pk["tp_w"] = torch.cat([sd["time_mlp.3.weight"], sd["prompt_mlp.weight"]], dim=1).to(f32).contiguous()
            pk["tp_b"] = (sd["time_mlp.3.bias"] + sd["prompt_mlp.bias"]).to(f32).contiguous()
        if ov_w:
            pk["ov_w"], pk["ov_b"] = torch.cat(ov_w, dim=0).contiguous(), torch.cat(ov_b, dim=0).contiguous()
        assert td == pk["t1_w"].shape[0]
        return pk

    # ------------------------------------------------------------------ buffers / small pieces
    def _buf(self, tag: str, rows: int, cols: int, dtype, dev) -> torch.Tensor:
        key = (tag, rows, cols, dtype)
        t = self._bufs.get(key)
        if t is None or t.device != dev:
            t = torch.empty(rows, cols, dtype=dtype, device=dev)
            self._bufs[key] = t
        return t

    def _time_embedding(self, time, B: int, dev) -> torch.Tensor:
        """SinusoidalPosEmb (module_util.py:36-48), evaluated on the host in the reference's fp32 op order for the
        (scalar) time and cached on the device; -> [B, nf]."""
        if torch.is_tensor(time):
            vals = [v for v in time.detach().reshape(-1).tolist()]
        else:
            vals = [time]
        if len(vals) not in (1, B):
            raise ValueError("time must be a scalar or have one entry per sample")
        key = (tuple(vals), B)
        emb = self._emb_cache.get(key)
        if emb is None or emb.device != dev:
            half = self.nf // 2
            e = math.log(10000) / (half - 1)
            freqs = torch.exp(torch.arange(half) * -e)
            t = torch.tensor(vals)
            arg = t[:, None] * freqs[None, :]
            emb = torch.cat((arg.sin(), arg.cos()), dim=-1).float()
            emb = emb.expand(B, -1).contiguous().to(dev)
            if len(self._emb_cache) > 4096:
                self._emb_cache.clear()
            self._emb_cache[key] = emb
        return emb

    def _resblock(self, r: dict, x: torch.Tensor, out: torch.Tensor, B: int, H: int, W: int, ss_all: torch.Tensor, tag: str,
                  precise_res: bool = False):
        """ResBlock (module_util.py:131-153): x [M, din] view -> out [M, dout] view (stream dtype).  ``precise_res``: the
        1x1 ``res_conv`` runs on fp32 rows with fp32 weights into an fp32 buffer (the precise tail path of bf16 mode)."""
        M, dev, adt = B * H * W, x.device, self._pack["adt"]
        din, dout, off = r["din"], r["dout"], r["off"]
        h = self._buf(tag + "h", M, dout, adt, dev)
        ops.conv_gemm(x, r["w1"], h, B=B, H=H, W=W, C_=din, N=dout, KH=3, KW=3, pad=1,
                      batch_scale=ss_all[:, off:off + dout], batch_bias=ss_all[:, off + dout:off + 2 * dout], act=ops.ACT_SILU)
        res = x
        if r["wres"] is not None and precise_res and x.dtype == torch.float32 and "wres32" in r:
            res = self._buf(tag + "r32", M, dout, torch.float32, dev)
            ops.conv_gemm(x, r["wres32"], res, B=B, H=H, W=W, C_=din, N=dout)
        elif r["wres"] is not None:
            res = self._buf(tag + "r", M, dout, adt, dev)
            ops.conv_gemm(x, r["wres"], res, B=B, H=H, W=W, C_=din, N=dout)
        ops.conv_gemm(h, r["w2"], out, B=B, H=H, W=W, C_=dout, N=dout, KH=3, KW=3, pad=1, act=ops.ACT_SILU, res=res)

    def _attn(self, a: dict, x: torch.Tensor, out: torch.Tensor, B: int, H: int, W: int, u_all, tag: str):
        """Residual(PreNorm(dim, LinearAttention | SpatialTransformer)): x [M, dim] view -> out [M, dim] view."""
        M, n, dev, adt, f32 = B * H * W, H * W, x.device, self._pack["adt"], torch.float32
        dim = a["g_pre"].numel()
        if a["kind"] == "linear" and adt == torch.bfloat16 and dim in (64, 128, 256) and self.fused_linattn:
            # the whole Residual(PreNorm(LinearAttention)) block in two passes over x (csrc/linattn.hip)
            ws = self._buf("laf_ws", 1, ops.linattn_block_workspace_bytes(B, n) // 4, f32, dev)
            ops.linattn_block(x, B, n, dim, a["g_pre"], a["w_qkv"], a["w_out"], a["b_out"], a["g_out"], out, ws)
            return
        if a["kind"] == "linear":
            xn = self._buf(tag + "xn", M, dim, adt, dev)
            ops.ln_rows(x, a["g_pre"], None, xn, 1e-5)
            qkv = self._buf(tag + "qkv", M, 384, adt, dev)
            ops.conv_gemm(xn, a["w_qkv"], qkv, B=B, H=H, W=W, C_=dim, N=384)
            la = self._buf(tag + "la", M, 128, adt, dev)
            ws = self._buf("la_ws", 1, ops.linattn_workspace_bytes(B, n, 4) // 4, f32, dev)
            ops.linear_attention(qkv, B, n, 4, la, ws)
            y = self._buf(tag + "y", M, dim, f32, dev)
            ops.conv_gemm(la, a["w_out"], y, B=B, H=H, W=W, C_=128, N=dim, bias=a["b_out"])
            if x.dtype == f32:
                ops.ln_rows(y, a["g_out"], None, out, 1e-5, res=x)
            else:   # bf16 stream: the norm's fp32 residual slot cannot take x
                ops.ln_rows(y, a["g_out"], None, y, 1e-5)
                ops.add_rows(y, x, out)
            return
        if u_all is None:
            raise ValueError("image_context is required: the model was built with use_image_context=True")
        heads = a["heads"]
        xn = self._buf(tag + "xn", M, dim, f32, dev)
        ops.ln_rows(x, a["g_pre"], None, xn, 1e-5)
        ga, gs = self._buf(tag + "ga", B, dim, f32, dev), self._buf(tag + "gs", B, dim, f32, dev)
        ws = self._buf("gn_ws", 1, max(ops.ln2d_workspace_bytes(B, n, dim) // 8, 1), torch.float64, dev)
        ops.groupnorm_affine(xn, B, n, dim, 32, 1e-6, a["gn_w"], a["gn_b"], ga, gs, ws)
        t0 = self._buf(tag + "t0", M, dim, f32, dev)
        ops.conv_gemm(xn, a["w_in"], t0, B=B, H=H, W=W, C_=dim, N=dim, a_scale=ga, a_shift=gs, bias=a["b_in"])
        h = self._buf(tag + "h", M, dim, adt, dev)
        ops.ln_rows(t0, a["n1"][0], a["n1"][1], h, 1e-5)
        qkv = self._buf(tag + "qkv", M, 3 * dim, adt, dev)
        ops.conv_gemm(h, a["w_qkv"], qkv, B=B, H=H, W=W, C_=dim, N=3 * dim, col_scale=a["qcs"])
        att = self._buf(tag + "att", M, dim, adt, dev)
        ops.flash_attn(qkv[:, :dim], qkv[:, dim:2 * dim], qkv[:, 2 * dim:], None, B, n, heads, 32, 0.0, att, None)
        t1 = self._buf(tag + "t1", M, dim, f32, dev)
        o = a["ov_off"]
        ops.conv_gemm(att, a["w_o1"], t1, B=B, H=H, W=W, C_=dim, N=dim, bias=a["b_o1"], batch_bias=u_all[:, o:o + dim], res=t0)
        ops.ln_rows(t1, a["n3"][0], a["n3"][1], h, 1e-5)
        gg = self._buf(tag + "gg", M, 4 * dim, adt, dev)     # GEGLU in the projection's epilogue (rows paired by _build_pack)
        ops.conv_gemm(h, a["w_ff0"], gg, B=B, H=H, W=W, C_=dim, N=8 * dim, bias=a["b_ff0"], store_mode=2)
        t2 = self._buf(tag + "t2", M, dim, adt, dev)     # feeds proj_out only: MFMA-operand dtype (bf16 rows take the LDS-DMA GEMM)
        ops.conv_gemm(gg, a["w_ff2"], t2, B=B, H=H, W=W, C_=4 * dim, N=dim, bias=a["b_ff2"], res=t1)
        y = self._buf(tag + "y", M, dim, f32, dev)
        ops.conv_gemm(t2, a["w_pout"], y, B=B, H=H, W=W, C_=dim, N=dim, bias=a["b_pout"], res=xn)      # ST: + x_in
        ops.add_rows(y, x, out)                                                                          # Residual: + x

    # ------------------------------------------------------------------ forward
    def _conditioning_rows(self, emb: torch.Tensor, R: int, B: int, text_context, image_context, dev, tag: str):
        """emb [R, nf] (R = B, or T * B for a schedule) -> (ss_all [R, S], u_all [B, U] or None, st [R, td])."""
        pk, f32, td = self._pack, torch.float32, 4 * self.nf
        tcat = self._buf(tag + "tcat", R, 2 * td, f32, dev)
        ops.linear(emb, pk["t1_w"], pk["t1_b"], tcat[:, :td], ops.ACT_GELU)
        st = self._buf(tag + "st", R, td, f32, dev)
        if self._has_prompt and text_context is not None:
            p1, p2 = self._buf("p1", B, td, f32, dev), self._buf("p2", B, td, f32, dev)
            ops.linear(text_context.contiguous().float(), pk["x0_w"], pk["x0_b"], p1, ops.ACT_SILU)
            ops.linear(p1, pk["x2_w"], pk["x2_b"], p2)
            if R == B:
                ops.softmax_rows(p2, pk["prompt"], tcat[:, td:])
            else:   # the prompt half does not depend on the time: one softmax, copied to every step's rows
                pr = self._buf("prow", B, td, f32, dev)
                ops.softmax_rows(p2, pk["prompt"], pr)
                tcat.view(R // B, B, 2 * td)[:, :, td:] = pr
            ops.linear(tcat, pk["tp_w"], pk["tp_b"], st, ops.ACT_SILU)
        else:
            ops.linear(tcat[:, :td], pk["t3_w"], pk["t3_b"], st, ops.ACT_SILU)
        ss_all = self._buf(tag + "ss_all", R, pk["ss_w"].shape[0], f32, dev)
        ops.linear(st, pk["ss_w"], pk["ss_b"], ss_all)
        u_all = None
        if "ov_w" in pk and self._has_st and image_context is not None:
            u_all = self._buf(tag + "u_all", B, pk["ov_w"].shape[0], f32, dev)
            ops.linear(image_context.contiguous().float(), pk["ov_w"], pk["ov_b"], u_all)
        return ss_all, u_all, st

    def _conditioning(self, time, B: int, text_context, image_context, dev, use_schedule: bool = True):
        sch = self._schedule if use_schedule else None
        if sch is not None and not torch.is_tensor(time):
            i = sch["index"].get(float(time))
            if (i is not None and sch["B"] == B and sch["dev"] == dev and sch["text"] is text_context
                    and sch["image"] is image_context and sch["pack"] is self._pack
                    and sch["versions"] == self._ctx_versions(text_context, image_context)):
                return sch["ss"][i], sch["u"]
        emb = self._time_embedding(time, B, dev)
        ss_all, u_all, _ = self._conditioning_rows(emb, B, B, text_context, image_context, dev, "")
        return ss_all, u_all

    @staticmethod
    def _ctx_versions(*ts):
        return tuple(None if t is None else t._version for t in ts)

    @contextlib.contextmanager
    def conditioning_schedule(self, times, B: int, text_context=None, image_context=None, device=None):
        """Everything the UNet derives from (time, text_context, image_context) alone -- the time MLP, the prompt
        softmax, every ResBlock's (scale + 1 | shift) vector, the cross-attention constants -- for ALL ``times`` of a
        sampling loop in one batched pass (6 tiny launches per step otherwise: 0.15 of 5.5 ms at B = 8, 256 x 256).
        Inside the ``with`` block, forward() calls with one of these scalar times, this batch size and these very
        context tensors take their rows from it; anything else is computed as usual -- including calls after an in-place
        edit of a context tensor (``tensor._version`` is recorded) or a weight reload (the pack's identity is recorded).  Row results of ``ops.linear`` do
        not depend on how many rows share a launch, so the outputs are bit-identical."""
        if self._pack is None:
            self._pack = self._build_pack()
        dev = device if device is not None else (text_context.device if text_context is not None else
                                                 image_context.device if image_context is not None else torch.device("cuda"))
        times = [float(t) for t in times]
        with torch.cuda.device(dev):
            emb = torch.cat([self._time_embedding(t, B, dev) for t in times], 0)
            ss, u, _ = self._conditioning_rows(emb, len(times) * B, B, text_context, image_context, dev, "sch_")
        prev = self._schedule
        if prev is not None:      # nested schedule: the "sch_" buffers are keyed by shape only, keep the outer one's rows
            ss, u = ss.clone(), (None if u is None else u.clone())
        self._schedule = dict(index={t: i for i, t in enumerate(times)}, B=B, dev=dev, text=text_context, image=image_context,
                              ss=ss.view(len(times), B, -1), u=u, pack=self._pack,
                              versions=self._ctx_versions(text_context, image_context))
        try:
            yield self
        finally:
            self._schedule = prev

    @torch.no_grad()
    @ops.on_operand_device
    def forward(self, xt: torch.Tensor, cond: torch.Tensor, time, text_context: Optional[torch.Tensor] = None,
                image_context: Optional[torch.Tensor] = None, taps: Optional[dict] = None) -> torch.Tensor:
        if not xt.is_cuda:
            raise RuntimeError("ConditionalUNet (MI355X build) runs on a ROCm device only; there is no CPU fallback")
        if xt.dim() != 4 or xt.shape != cond.shape or xt.shape[1] != self.in_nc:
            raise ValueError("xt and cond must both be (B, in_nc, H, W)")
        if self._pack is None:
            self._pack = self._build_pack()
        pk = self._pack
        dev, f32 = xt.device, torch.float32
        B, _, H, W = xt.shape
        s = 2 ** self.depth
        Hp, Wp = H + (s - H % s) % s, W + (s - W % s) % s
        if Hp - H >= H or Wp - W >= W:
            raise ValueError("image too small for reflect padding")
        nf, td, lv, D = self.nf, 4 * self.nf, self._levels(), self.depth
        xt, cond = xt.contiguous().float(), cond.contiguous().float()

        # ---- time / prompt embedding -> SiLU(t) -> all ResBlock (scale+1 | shift) vectors in one linear (or this
        # step's rows of a schedule computed once for the whole sampling loop: conditioning_schedule)
        ss_all, u_all = self._conditioning(time, B, text_context, image_context, dev, use_schedule=taps is None)
        st = self._bufs.get(("st", B, td, f32)) if taps is not None else None

        # ---- buffers: concat buffers per level ([x | skip]); producers write into their column slices
        Hs = [Hp >> i for i in range(D)]
        Ws = [Wp >> i for i in range(D)]
        Ms = [B * Hs[i] * Ws[i] for i in range(D)]
        sdt = pk["adt"]    # the activation stream's dtype: fp32, or bf16 in bf16 mode
        # precise tail path (bf16 mode): the full-resolution tensors the noise prediction is a short linear map of -- the
        # stem features and the last up conv's output (the two halves of catF), the final ResBlock's res_conv and output
        # -- stay fp32 and that res_conv is an fp32 GEMM: a bf16 step's error against the fp32 model falls from 5.9e-3
        # to ~3.4e-3 rel. L2 (tools/exp/sde_err_sets.py), which is what a 100-step sampling loop needs to hold 50 dB
        tdt = f32 if (self.precise_tail or sdt == f32) else sdt
        catA = [self._buf(f"catA{i}", Ms[i], lv[i][1] + lv[i][0], sdt, dev) for i in range(D)]
        catB = [self._buf(f"catB{i}", Ms[i], lv[i][1] + lv[i][0], sdt, dev) for i in range(D)]
        catF = self._buf("catF", Ms[0], 2 * nf, tdt, dev)

        x_in = self._buf("x_in", Ms[0], 8, f32, dev)
        ops.unet_input(xt, cond, Hp, Wp, x_in)
        ops.conv_gemm(x_in, pk["init_w"], catF[:, nf:], B=B, H=Hp, W=Wp, C_=8, N=nf, KH=7, KW=7, pad=3)
        if taps is not None:
            taps["st"], taps["init"] = st.clone(), self._nchw(catF[:, nf:], B, Hp, Wp)

        x = catF[:, nf:]
        for i in range(D):
            din, dout = lv[i]
            d = pk["downs"][i]
            h1, h2 = catB[i][:, dout:], catA[i][:, dout:]
            self._resblock(d["b1"], x, h1, B, Hs[i], Ws[i], ss_all, f"d{i}a")
            tmp = self._buf(f"d{i}t", Ms[i], din, sdt, dev)
            self._resblock(d["b2"], h1, tmp, B, Hs[i], Ws[i], ss_all, f"d{i}b")
            self._attn(d["attn"], tmp, h2, B, Hs[i], Ws[i], u_all, f"d{i}c")
            if taps is not None:
                taps[f"down{i}"] = self._nchw(h2, B, Hs[i], Ws[i])
            if i != D - 1:
                nxt = self._buf(f"xin{i + 1}", Ms[i + 1], dout, sdt, dev)
                ops.conv_gemm(h2, d["w_down"], nxt, B=B, H=Hs[i], W=Ws[i], C_=din, N=dout, KH=4, KW=4, stride=2, pad=1,
                              bias=d["b_down"])
            else:
                nxt = self._buf("xmid", Ms[i], dout, sdt, dev)
                ops.conv_gemm(h2, d["w_down"], nxt, B=B, H=Hs[i], W=Ws[i], C_=din, N=dout, KH=3, KW=3, pad=1)
            x = nxt
        Hm, Wm, Mm, dm = Hs[-1], Ws[-1], Ms[-1], lv[-1][1]
        m1, m2 = self._buf("m1", Mm, dm, sdt, dev), self._buf("m2", Mm, dm, sdt, dev)
        self._resblock(pk["mid1"], x, m1, B, Hm, Wm, ss_all, "m1")
        self._attn(pk["mid_attn"], m1, m2, B, Hm, Wm, u_all, "ma")
        self._resblock(pk["mid2"], m2, catA[D - 1][:, :dm], B, Hm, Wm, ss_all, "m2")
        if taps is not None:
            taps["mid"] = self._nchw(catA[D - 1][:, :dm], B, Hm, Wm)
        for j in range(D):
            i = D - 1 - j
            din, dout = lv[i]
            u = pk["ups"][j]
            self._resblock(u["b1"], catA[i], catB[i][:, :dout], B, Hs[i], Ws[i], ss_all, f"u{j}a")
            t2, t3 = self._buf(f"u{j}t2", Ms[i], dout, sdt, dev), self._buf(f"u{j}t3", Ms[i], dout, sdt, dev)
            self._resblock(u["b2"], catB[i], t2, B, Hs[i], Ws[i], ss_all, f"u{j}b")
            self._attn(u["attn"], t2, t3, B, Hs[i], Ws[i], u_all, f"u{j}c")
            if taps is not None:
                taps[f"up{j}"] = self._nchw(t3, B, Hs[i], Ws[i])
            if i != 0:
                ops.conv_gemm(t3, u["w_up"], catA[i - 1][:, :din], B=B, H=Hs[i], W=Ws[i], C_=dout, N=din, KH=3, KW=3, pad=1,
                              ups=1, bias=u["b_up"])
            else:
                ops.conv_gemm(t3, u["w_up"], catF[:, :nf], B=B, H=Hs[i], W=Ws[i], C_=dout, N=din, KH=3, KW=3, pad=1)
        fin = self._buf("fin", Ms[0], nf, tdt, dev)
        self._resblock(pk["final"], catF, fin, B, Hp, Wp, ss_all, "fin", precise_res=tdt != sdt)
        out = torch.empty(B, self.out_nc, H, W, dtype=f32, device=dev)
        ops.tail_conv3x3(fin, B, Hp, Wp, nf, pk["tail_w"], pk["tail_b"], self.out_nc, out, split=tdt != sdt)
        return out

    @staticmethod
    def _nchw(t: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        return t.reshape(B, H, W, -1).permute(0, 3, 1, 2).float().clone()
