"""``DaCLIPVision`` -- the image side of the reference's ``DaCLIP`` (two ViT towers) on gfx950 kernels.

Mirrors /root/reference/legacy/third_party/universal-image-restoration/open_clip/daclip_model.py:17-55:
  * ``encode_image(image, control=False, normalize=False)`` -> ``image_features`` or, with ``control=True``,
    ``(image_features, degra_features)`` (both (B, output_dim)); ``degra_features`` is the ``z_d`` /
    ``text_context`` the restoration nets are conditioned on (U/test.py:114-117).
  * ``visual_control(image, output_hiddens=True)`` -> ``(degra_features, hiddens)`` and
    ``visual(image, control=hiddens)`` as separate calls (teacher/export_pseudolabels.py:274), hiddens in the
    reference's LND layout; the list passed as ``control`` is consumed from the end like the reference's
    ``control.pop()`` (open_clip/transformer.py:359-368).
  * ``state_dict`` keys / shapes of ``visual.*`` and ``visual_control.*`` as in the reference
    (synth.daclip_vision_param_shapes; ``clip.visual.*`` aliases in a checkpoint are accepted).
The text tower / tokenizer are out of scope (prompts are encoded offline), see DESIGN.md.

Tokens are rows of a ``[B*L, width]`` matrix (L = patches + 1), residual stream fp32.  Per layer: one LayerNorm
kernel, QKV GEMM (Q leaves pre-scaled by log2(e)/sqrt(hd)), flash attention, out-proj GEMM (+residual), LayerNorm,
FC GEMM (+exact GELU), proj GEMM (+residual).  With ``control=True`` the control tower keeps its 12 residual
streams and the main tower applies ``x += zero_linear_j(xc_j)`` as ONE GEMM accumulating onto its stream, so the
hiddens are never materialised.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .student import _Holder, _attach
from .synth import daclip_vision_param_shapes

_LOG2E = 1.4426950408889634
_EPS = 1e-5


class _Tower(_Holder):
    """``model.visual`` / ``model.visual_control``: parameter holder that is also callable like the reference's
    ``VisionTransformer.forward(x, output_hiddens=False, control=None)`` (open_clip/transformer.py:507)."""

    def forward(self, image, output_hiddens: bool = False, control: Optional[list] = None):
        owner, which = self.__dict__["_owner"][0], self.__dict__["_which"]
        return owner._tower_call(which, image, output_hiddens, control)


class DaCLIPVision(nn.Module):
    def __init__(self, image_size: int = 224, patch_size: int = 32, width: int = 768, layers: int = 12,
                 heads: Optional[int] = None, output_dim: int = 512, precision: str = "fp32") -> None:
        super().__init__()
        heads = heads if heads is not None else width // 64   # open_clip/model.py:123 (head_width 64)
        if width % heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        hd = width // heads
        if hd % 4 or hd > 128:
            raise ValueError("this build needs head_dim % 4 == 0 and head_dim <= 128")
        if patch_size % 4 or image_size % patch_size or width % 8:
            raise ValueError("this build needs patch_size % 4 == 0, image_size % patch_size == 0, width % 8 == 0")
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.image_size, self.patch_size, self.width, self.layers = image_size, patch_size, width, layers
        self.heads, self.output_dim, self.precision = heads, output_dim, precision
        self.grid = image_size // patch_size
        self.visual, self.visual_control = _Tower(), _Tower()
        for t, which in ((self.visual, "visual"), (self.visual_control, "visual_control")):
            t.__dict__["_owner"], t.__dict__["_which"] = [self], which
        for key, shape in daclip_vision_param_shapes(image_size, patch_size, width, layers, output_dim).items():
            _attach(self, key, shape)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if ".ln_" in name and name.endswith("weight"):
                    p.fill_(1.0)
                elif name.endswith("bias") or "zero_modules" in name:
                    p.zero_()   # zero_module(...) init of the control links, transformer.py:295-306
                else:
                    p.normal_(0.0, width ** -0.5)
        self._pack: Optional[dict] = None
        self._bufs: Dict[tuple, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    def set_precision(self, precision: str) -> "DaCLIPVision":
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if precision != self.precision:
            self.precision, self._pack = precision, None
        return self

    def _apply(self, fn, *a, **k):
        self._pack, self._bufs = None, {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._pack = None
        sd = dict(state_dict)
        for k in list(sd):  # a full DaCLIP checkpoint holds the CLIP ViT twice (clip.visual.* is visual.*)
            if k.startswith("clip.visual."):
                sd.setdefault(k[len("clip."):], sd[k])
        if not strict:
            own = set(self.state_dict().keys())
            sd = {k: v for k, v in sd.items() if k in own}
        return super().load_state_dict(sd, strict=strict, **kw)

    def _build_pack(self) -> dict:
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        wdt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        f32 = torch.float32
        W, hd = self.width, self.width // self.heads
        dev = sd["visual.proj"].device
        cs = torch.ones(3 * W, device=dev, dtype=f32)
        cs[:W] = _LOG2E / math.sqrt(hd)
        pk: dict = {"adt": wdt, "qcs": cs}
        for which, tp in (("visual", "visual.transformer."), ("visual_control", "visual_control.transformer.transformer.")):
            p = which + "."
            t: dict = {}
            t["conv_w"] = sd[p + "conv1.weight"].reshape(W, -1).to(wdt).contiguous()
            posc = sd[p + "positional_embedding"].to(f32).clone()
            posc[0] += sd[p + "class_embedding"].to(f32)  # row 0 of the patch GEMM is exactly 0 (zero patch row)
            t["posc"] = posc.contiguous()
            t["ln_pre"] = (sd[p + "ln_pre.weight"].to(f32).contiguous(), sd[p + "ln_pre.bias"].to(f32).contiguous())
            t["ln_post"] = (sd[p + "ln_post.weight"].to(f32).contiguous(), sd[p + "ln_post.bias"].to(f32).contiguous())
            t["proj_t"] = sd[p + "proj"].t().to(f32).contiguous()
            t["blocks"] = []
            for i in range(self.layers):
                r = f"{tp}resblocks.{i}."
                g = lambda k: sd[r + k].to(f32).contiguous()  # noqa: E731
                m = lambda k: sd[r + k].to(wdt).contiguous()  # noqa: E731
                t["blocks"].append(dict(
                    ln1=(g("ln_1.weight"), g("ln_1.bias")), ln2=(g("ln_2.weight"), g("ln_2.bias")),
                    qkv_w=m("attn.in_proj_weight"), qkv_b=g("attn.in_proj_bias"),
                    out_w=m("attn.out_proj.weight"), out_b=g("attn.out_proj.bias"),
                    fc_w=m("mlp.c_fc.weight"), fc_b=g("mlp.c_fc.bias"),
                    pj_w=m("mlp.c_proj.weight"), pj_b=g("mlp.c_proj.bias")))
            pk[which] = t
        pk["zero"] = [(sd[f"visual_control.transformer.zero_modules.{i}.weight"].to(wdt).contiguous(),
                       sd[f"visual_control.transformer.zero_modules.{i}.bias"].to(f32).contiguous())
                      for i in range(self.layers)]
        return pk

    def _buf(self, tag: str, rows: int, cols: int, dtype, dev) -> torch.Tensor:
        key = (tag, rows, cols, dtype)
        t = self._bufs.get(key)
        if t is None or t.device != dev:
            t = torch.empty(rows, cols, dtype=dtype, device=dev)
            self._bufs[key] = t
        return t

    # ------------------------------------------------------------------ building blocks
    def _check_image(self, image: torch.Tensor) -> torch.Tensor:
        if not image.is_cuda:
            raise RuntimeError("DaCLIPVision (MI355X build) runs on a ROCm device only; there is no CPU fallback")
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != self.image_size or image.shape[3] != self.image_size:
            raise ValueError(f"image must be (B, 3, {self.image_size}, {self.image_size})")
        if self._pack is None:
            self._pack = self._build_pack()
        return image.contiguous().float()

    def _patches(self, image: torch.Tensor) -> torch.Tensor:
        B, L = image.shape[0], self.grid * self.grid + 1
        a = self._buf("patches", B * L, 3 * self.patch_size ** 2, self._pack["adt"], image.device)
        ops.patchify(image, self.patch_size, a)
        return a

    def _embed(self, t: dict, patches: torch.Tensor, x: torch.Tensor) -> None:
        """conv1 + class/positional embedding + ln_pre (transformer.py:522-536) -> x [B*L, W] fp32."""
        tmp = self._buf("embed", x.shape[0], self.width, torch.float32, x.device)
        ops.gemm(patches, t["conv_w"], tmp)
        ops.ln_rows(tmp, t["ln_pre"][0], t["ln_pre"][1], x, _EPS, add=t["posc"], add_period=self.grid * self.grid + 1)

    def _resblock(self, bp: dict, x_in: torch.Tensor, x_out: torch.Tensor, B: int) -> None:
        """ResidualAttentionBlock (transformer.py:232-244): x_out = y + mlp(ln_2(y)), y = x_in + attn(ln_1(x_in)).
        x_in is left intact when x_out is another buffer."""
        M, W = x_in.shape
        L = M // B
        dev, adt = x_in.device, self._pack["adt"]
        h = self._buf("h", M, W, adt, dev)
        ops.ln_rows(x_in, bp["ln1"][0], bp["ln1"][1], h, _EPS)
        qkv = self._buf("qkv", M, 3 * W, adt, dev)
        ops.gemm(h, bp["qkv_w"], qkv, bias=bp["qkv_b"], col_scale=self._pack["qcs"])
        att = self._buf("att", M, W, adt, dev)
        ops.flash_attn(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], None, B, L, self.heads, W // self.heads, 0.0, att, None)
        y = self._buf("y", M, W, torch.float32, dev)
        ops.gemm(att, bp["out_w"], y, bias=bp["out_b"], res=x_in)
        ops.ln_rows(y, bp["ln2"][0], bp["ln2"][1], h, _EPS)
        f = self._buf("fc", M, 4 * W, adt, dev)
        ops.gemm(h, bp["fc_w"], f, bias=bp["fc_b"], act=ops.ACT_GELU)
        ops.gemm(f, bp["pj_w"], x_out, bias=bp["pj_b"], res=y)

    def _pool(self, t: dict, x: torch.Tensor, B: int) -> torch.Tensor:
        """CLS token -> ln_post -> @ proj (transformer.py:548-552)."""
        L, W = x.shape[0] // B, self.width
        cls = x.view(B, L * W)[:, :W]
        pooled = self._buf("pooled", B, W, torch.float32, x.device)
        ops.ln_rows(cls, t["ln_post"][0], t["ln_post"][1], pooled, _EPS)
        out = torch.empty(B, self.output_dim, dtype=torch.float32, device=x.device)
        ops.linear(pooled, t["proj_t"], None, out)
        return out

    def _control_streams(self, patches: torch.Tensor, B: int) -> List[torch.Tensor]:
        """Control tower: residual stream after every resblock (xc[i]); hiddens[i] = zero_linear_i(xc[i])."""
        t = self._pack["visual_control"]
        M, dev = patches.shape[0], patches.device
        x0 = self._buf("xc_in", M, self.width, torch.float32, dev)
        self._embed(t, patches, x0)
        xs, cur = [], x0
        for i, bp in enumerate(t["blocks"]):
            nxt = self._buf(f"xc{i}", M, self.width, torch.float32, dev)
            self._resblock(bp, cur, nxt, B)
            xs.append(nxt)
            cur = nxt
        return xs

    def _main(self, patches: torch.Tensor, B: int, streams: Optional[List[torch.Tensor]],
              hiddens: Optional[List[torch.Tensor]]) -> torch.Tensor:
        t = self._pack["visual"]
        M, dev = patches.shape[0], patches.device
        x = self._buf("xm", M, self.width, torch.float32, dev)
        self._embed(t, patches, x)
        for i, bp in enumerate(t["blocks"]):
            self._resblock(bp, x, x, B)
            if streams is not None:      # x += zero_linear_j(xc_j), j = L-1-i  (control.pop(): last first)
                j = self.layers - 1 - i
                wz, bz = self._pack["zero"][j]
                ops.gemm(streams[j], wz, x, bias=bz, res=x)
            elif hiddens is not None:    # explicit hiddens handed in by the caller (consumed from the end, LND)
                hrows = hiddens.pop().permute(1, 0, 2).reshape(M, self.width).float().contiguous()
                ops.add_rows(x, hrows, x)
        return self._pool(t, x, B)

    # ------------------------------------------------------------------ reference-shaped API
    @torch.no_grad()
    @ops.on_operand_device
    def encode_image(self, image: torch.Tensor, control: bool = False, normalize: bool = False):
        image = self._check_image(image)
        B = image.shape[0]
        patches = self._patches(image)
        if not control:
            f = self._main(patches, B, None, None)
            return F.normalize(f, dim=-1) if normalize else f
        streams = self._control_streams(patches, B)
        degra = self._pool(self._pack["visual_control"], streams[-1], B)
        img = self._main(patches, B, streams, None)
        if normalize:
            img, degra = F.normalize(img, dim=-1), F.normalize(degra, dim=-1)
        return img, degra

    @torch.no_grad()
    def _tower_call(self, which: str, image: torch.Tensor, output_hiddens: bool, control: Optional[list]):
        if image.is_cuda and image.device.index != torch.cuda.current_device():
            with torch.cuda.device(image.device):
                return self._tower_call(which, image, output_hiddens, control)
        image = self._check_image(image)
        B = image.shape[0]
        patches = self._patches(image)
        L = self.grid * self.grid + 1
        if which == "visual_control":
            if control is not None:
                raise NotImplementedError("the control tower takes no control input in DA-CLIP (daclip_model.py:48)")
            streams = self._control_streams(patches, B)
            degra = self._pool(self._pack["visual_control"], streams[-1], B)
            if not output_hiddens:
                return degra
            hiddens = []
            for j, xc in enumerate(streams):
                wz, bz = self._pack["zero"][j]
                hz = torch.empty(B * L, self.width, dtype=torch.float32, device=image.device)
                ops.gemm(xc, wz, hz, bias=bz)
                hiddens.append(hz.view(B, L, self.width).permute(1, 0, 2))  # LND like the reference
            return degra, hiddens
        if output_hiddens:
            raise NotImplementedError("output_hiddens is only used on the control tower (daclip_model.py:48)")
        return self._main(patches, B, None, control)

    @torch.no_grad()
    def project_tokens(self, which: str, hidden: torch.Tensor) -> torch.Tensor:
        """(L, B, width) tokens of tower `which` ("visual" / "control") -> (B, output_dim, g, g): the tower's ln_post and
        projection (what transformer.py:548-552 applies to the pooled class token) applied to every PATCH token, so
        that dense tokens can be compared with the prompt embeddings (teacher.daclip_patch_tokens)."""
        if self._pack is None:
            self._pack = self._build_pack()
        t = self._pack["visual_control" if which in ("control", "visual_control") else "visual"]
        L, B, W = hidden.shape
        g = self.grid
        if L != g * g + 1 or W != self.width or not hidden.is_cuda:
            raise ValueError(f"hidden must be a ROCm tensor of shape ({g * g + 1}, B, {self.width})")
        rows = hidden[1:].permute(1, 0, 2).reshape(B * g * g, W).float().contiguous()
        normed = torch.empty_like(rows)
        ops.ln_rows(rows, t["ln_post"][0], t["ln_post"][1], normed, _EPS)
        out = torch.empty(B * g * g, self.output_dim, dtype=torch.float32, device=hidden.device)
        ops.linear(normed, t["proj_t"], None, out)
        return out.view(B, g, g, self.output_dim).permute(0, 3, 1, 2)

    def forward(self, image: torch.Tensor, control: bool = True, normalize: bool = False):
        return self.encode_image(image, control=control, normalize=normalize)
