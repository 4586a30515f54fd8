"""Thin Python launchers over the C ABI (include/uwir.h).  Tensors are CUDA(HIP) torch tensors
used purely as device-memory handles: 2-D ``[rows, ld]`` views (NHWC activations) whose
``data_ptr()`` / ``stride(0)`` are handed to the kernels.  Nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, BF16, F32, GemmArgs  # noqa: F401


def _stream(t) -> int:
    """HIP stream handle every launch of an op goes to: torch's current stream of the CURRENT device.  `t` is an
    operand (tensor or device) of the op: the library launches in the current HIP device context, so an operand on
    another GPU is an error here (it used to enqueue on the wrong GPU's stream silently).  The module-level entry
    points (student / UNet / DA-CLIP / sampler forwards) make their input's device current themselves."""
    s = torch.cuda.current_stream()
    dev = t.device if isinstance(t, torch.Tensor) else t
    if dev.type != "cuda" or (dev.index is not None and dev.index != s.device_index):
        raise ValueError(f"operand on {dev} but the current HIP device is cuda:{s.device_index}: run under "
                         f"torch.cuda.device(...) (the model-level forwards do)")
    return s.cuda_stream


def on_operand_device(fn):
    """Decorator of the model-level entry points (methods whose first argument is a tensor): run with that tensor's
    GPU as the current HIP device, so a model living on a non-current GPU launches in its own context / stream."""
    @functools.wraps(fn)
    def wrapper(self, x, *a, **k):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return fn(self, x, *a, **k)
        return fn(self, x, *a, **k)
    return wrapper


# Optional HIP-event timing of selected launches (bench.py's roofline leg).  Events are recorded on
# torch's current stream, which is the stream every kernel of this library is enqueued on.
_TIMING = {"on": False, "events": [], "all": False, "ops": []}


def event_timing(enable: bool) -> None:
    _TIMING["on"] = enable
    _TIMING["events"] = []


def op_census(enable: bool) -> None:
    """Per-call HIP events around EVERY launcher below (bench.py's per-kernel-class roofline list).  Meant for one
    untimed pass on a single stream: the events bracket whatever the call enqueues (for flash_attn that includes its
    pack pre-pass), plus a few microseconds of launch gap."""
    _TIMING["all"] = enable
    _TIMING["ops"] = []


def collect_op_census():
    """-> list of (op name, args, kwargs, milliseconds); call after torch.cuda.synchronize()."""
    out = [(n, a, k, s.elapsed_time(e)) for n, a, k, s, e in _TIMING["ops"]]
    _TIMING["ops"] = []
    return out


def collect_event_timings():
    """-> list of (tag, milliseconds); call after torch.cuda.synchronize()."""
    out = [(tag, s.elapsed_time(e)) for tag, s, e in _TIMING["events"]]
    _TIMING["events"] = []
    return out


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _f32(*ts) -> None:
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError("expected a float32 tensor")


def _ld(t: torch.Tensor) -> int:
    """Row stride of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("expected a 2-D view with unit inner stride")
    return t.stride(0)


def dwt2(x: torch.Tensor, B: int, H: int, W: int, C: int, ll: torch.Tensor, hf: torch.Tensor) -> None:
    lib = _lib.load()
    if _dt(ll) != _dt(hf):
        raise TypeError("dwt2: ll and hf must share a dtype")
    _lib.check(lib.uwir_dwt2(_p(x), _dt(x), _ld(x), B, H, W, C, _p(ll), _ld(ll), _p(hf), _ld(hf), _dt(ll), _stream(x)), "dwt2")


def idwt2(ll: torch.Tensor, hf: torch.Tensor, B: int, h: int, w: int, C: int, out: torch.Tensor) -> None:
    _f32(ll)
    lib = _lib.load()
    _lib.check(lib.uwir_idwt2(_p(ll), _ld(ll), _p(hf), _dt(hf), _ld(hf), B, h, w, C, _p(out), _dt(out), _ld(out), _stream(ll)),
               "idwt2")


def gate_idwt_supported(C: int, Kg: int) -> bool:
    return bool(_lib.load().uwir_gate_idwt_supported(C, Kg))


def gate_idwt(gate_in: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, bias1: torch.Tensor, b2: torch.Tensor,
              hf: torch.Tensor, B: int, h: int, w: int, C: int, out: torch.Tensor, alpha: torch.Tensor,
              merged: Optional[torch.Tensor] = None) -> None:
    """WFIGate (wfi_gate.py:29-45) + idwt2 in one launch: see include/uwir.h.  bf16 weights / hf / out, fp32 gate rows."""
    _f32(gate_in, bias1, b2, alpha, merged)
    bf = torch.bfloat16
    if w1.dtype != bf or w2.dtype != bf or hf.dtype != bf or out.dtype != bf:
        raise ValueError("gate_idwt: w1, w2, hf and out must be bf16")
    Kg = gate_in.shape[1]
    if tuple(w1.shape) != (64, Kg) or tuple(w2.shape) != (C, 64) or not w1.is_contiguous() or not w2.is_contiguous():
        raise ValueError("gate_idwt: w1 must be [64, Kg] and w2 [C, 64], contiguous")
    M = B * h * w
    if gate_in.shape[0] != M or hf.shape[0] != M or hf.shape[1] < 3 * C or out.shape[0] != 4 * M or out.shape[1] < C:
        raise ValueError("gate_idwt: row counts do not match B * h * w")
    if alpha.numel() != M * C or not alpha.is_contiguous() or tuple(bias1.shape) != (B, 64) or not bias1.is_contiguous():
        raise ValueError("gate_idwt: alpha must be a contiguous [B, h, w, C] and bias1 a contiguous [B, 64]")
    if merged is not None and (tuple(merged.shape) != (M, C) or not merged.is_contiguous()):
        raise ValueError("gate_idwt: merged must be a contiguous [M, C]")
    lib = _lib.load()
    _lib.check(lib.uwir_gate_idwt(_p(gate_in), _ld(gate_in), Kg, _p(w1), _p(w2), _p(bias1), _p(b2), _p(hf), _ld(hf), B, h, w, C,
                                  _p(out), _ld(out), _p(alpha), _p(merged), _stream(gate_in)), "gate_idwt")


def gate_blend(gate_in: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, bias1: torch.Tensor, b2: torch.Tensor,
               B: int, h: int, w: int, C: int, alpha: torch.Tensor, merged: torch.Tensor) -> None:
    """WFIGate without idwt2 (a block followed by another block of its stage): alpha fp32, merged as bf16 rows."""
    _f32(gate_in, bias1, b2, alpha)
    bf = torch.bfloat16
    if w1.dtype != bf or w2.dtype != bf or merged.dtype != bf:
        raise ValueError("gate_blend: w1, w2 and merged must be bf16")
    Kg = gate_in.shape[1]
    if tuple(w1.shape) != (64, Kg) or tuple(w2.shape) != (C, 64) or not w1.is_contiguous() or not w2.is_contiguous():
        raise ValueError("gate_blend: w1 must be [64, Kg] and w2 [C, 64], contiguous")
    M = B * h * w
    if gate_in.shape[0] != M or merged.shape[0] != M or merged.shape[1] < C:
        raise ValueError("gate_blend: row counts do not match B * h * w")
    if alpha.numel() != M * C or not alpha.is_contiguous() or tuple(bias1.shape) != (B, 64) or not bias1.is_contiguous():
        raise ValueError("gate_blend: alpha must be a contiguous [B, h, w, C] and bias1 a contiguous [B, 64]")
    lib = _lib.load()
    _lib.check(lib.uwir_gate_blend(_p(gate_in), _ld(gate_in), Kg, _p(w1), _p(w2), _p(bias1), _p(b2), B, h, w, C, _p(alpha),
                                   _p(merged), _ld(merged), _stream(gate_in)), "gate_blend")


def ffn_fused_supported(C: int, n: int) -> bool:
    return bool(_lib.load().uwir_ffn_fused_supported(C, n))


def ffn_permute_w2(w2: torch.Tensor) -> torch.Tensor:
    """ffn.2.weight [C, 2C] with the columns of every 32-block in the order uwir_ffn_fused reads them (include/uwir.h)."""
    K = w2.shape[1]
    pos = torch.arange(K, device=w2.device)
    s, q, e = pos // 32, (pos // 8) % 4, pos % 8
    src = 32 * s + torch.where(e < 4, 4 * q + e, 16 + 4 * q + e - 4)
    return w2[:, src].contiguous()


def ffn_fused(x: torch.Tensor, w0b: torch.Tensor, bias0b: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor, B: int, n: int,
              C: int, out: torch.Tensor) -> None:
    """out = W2 gelu(W0_b x + bias0_b) + b2 + x in one launch (wtb.py:117-124, bf16 mode): include/uwir.h."""
    _f32(bias0b, b2)
    bf = torch.bfloat16
    if x.dtype != bf or w0b.dtype != bf or w2p.dtype != bf or out.dtype != bf:
        raise ValueError("ffn_fused: x, w0b, w2p and out must be bf16")
    if tuple(w0b.shape) != (B, 2 * C, C) or w0b.stride(2) != 1 or w0b.stride(1) != C:
        raise ValueError("ffn_fused: w0b must be [B, 2C, C] with dense panels")
    if tuple(w2p.shape) != (C, 2 * C) or not w2p.is_contiguous() or tuple(bias0b.shape) != (B, 2 * C) or not bias0b.is_contiguous():
        raise ValueError("ffn_fused: w2p must be a contiguous [C, 2C] and bias0b a contiguous [B, 2C]")
    if x.shape[0] != B * n or out.shape[0] != B * n or x.shape[1] < C or out.shape[1] < C:
        raise ValueError("ffn_fused: x and out must have B * n rows of at least C channels")
    lib = _lib.load()
    _lib.check(lib.uwir_ffn_fused(_p(x), _ld(x), _p(w0b), w0b.stride(0), _p(bias0b), _p(w2p), _p(b2), B, n, C, _p(out), _ld(out),
                                  _stream(x)), "ffn_fused")


def ln2d_workspace_bytes(B: int, n: int, C_: int) -> int:
    return int(_lib.load().uwir_ln2d_workspace_bytes(B, n, C_))


def ln2d_affine(x: torch.Tensor, B: int, n: int, C_: int, eps: float, gamma, beta, scale_shift, a_out, s_out,
                mean_out, pooled_out, workspace: torch.Tensor) -> None:
    _f32(gamma, beta, scale_shift, a_out, s_out, mean_out, pooled_out)
    lib = _lib.load()
    ss_ld = 0 if scale_shift is None else _ld(scale_shift)
    _lib.check(lib.uwir_ln2d_affine(_p(x), _dt(x), _ld(x), B, n, C_, eps, _p(gamma), _p(beta), _p(scale_shift), ss_ld,
                                    _p(a_out), _p(s_out), _p(mean_out), _p(pooled_out), _p(workspace),
                                    workspace.numel() * workspace.element_size(), _stream(x)), "ln2d_affine")


def chan_attn_fold(a, s, w, out_scale, out_shift) -> None:
    _f32(a, s, w, out_scale, out_shift)
    lib = _lib.load()
    _lib.check(lib.uwir_chan_attn_fold(_p(a), _p(s), _p(w), _p(out_scale), _p(out_shift), a.numel(), _stream(a)),
               "chan_attn_fold")


def linear(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], y: torch.Tensor, act: int = ACT_NONE) -> None:
    _f32(x, W, bias, y)
    lib = _lib.load()
    B, I = x.shape
    O = W.shape[0]
    if W.shape[1] != I or not W.is_contiguous() or y.shape[0] != B or y.shape[1] != O:
        raise ValueError("linear: shape mismatch")
    _lib.check(lib.uwir_linear(_p(x), _ld(x), B, I, _p(W), _p(bias), O, act, _p(y), _ld(y), _stream(x)), "linear")


_GEMM_WS = {}    # split-K scratch per (device, stream): calls that share one must be ordered on one stream
                 # (uwir_gemm_args.workspace), so every stream that runs this library gets its own
_SPLITK = True   # tools/: False = never lend the scratch


def conv_gemm(A: torch.Tensor, Wt: torch.Tensor, out: Optional[torch.Tensor], *, B: int, H: int, W: int, C_: int,
              N: int, KH: int = 1, KW: int = 1, stride: int = 1, pad: int = 0, OH: Optional[int] = None,
              OW: Optional[int] = None, a_scale=None, a_shift=None, bias=None, batch_bias=None, act: int = ACT_NONE,
              res=None, res_scale=None, res_shift=None, store_mode: int = 0, blend_t=None, blend_f=None,
              blend_out=None, col_scale=None, batch_scale=None, ups: int = 0, w_batched: bool = False) -> None:
    """out[m,n] = epilogue(sum_k A[m,k] Wt[n,k]); see ``uwir_gemm_args`` in include/uwir.h.  ``w_batched``: Wt is
    [B, N, K], one weight panel per image (fold_affine_weights)."""
    lib = _lib.load()
    _f32(a_scale, a_shift, bias, batch_bias, res_scale, res_shift, blend_t, blend_f, blend_out, col_scale, batch_scale)
    g = GemmArgs()
    g.A, g.a_dtype, g.lda = _p(A), _dt(A), _ld(A)
    g.B, g.H, g.W, g.C = B, H, W, C_
    g.KH, g.KW, g.stride, g.pad = KH, KW, stride, pad
    g.OH = OH if OH is not None else ((H << ups) + 2 * pad - KH) // stride + 1
    g.OW = OW if OW is not None else ((W << ups) + 2 * pad - KW) // stride + 1
    g.a_scale, g.a_shift = _p(a_scale), _p(a_shift)
    if w_batched:
        if Wt.dim() != 3 or not Wt.is_contiguous() or tuple(Wt.shape) != (B, N, KH * KW * C_):
            raise ValueError(f"conv_gemm: batched weights must be contiguous [B, N, KH*KW*C], got {tuple(Wt.shape)}")
        g.w_batch_stride = N * KH * KW * C_
    elif Wt.dim() != 2 or not Wt.is_contiguous() or Wt.shape[0] != N or Wt.shape[1] != KH * KW * C_:
        raise ValueError(f"conv_gemm: weight must be contiguous [N, KH*KW*C], got {tuple(Wt.shape)}")
    g.Wt, g.w_dtype, g.N = _p(Wt), _dt(Wt), N
    g.bias, g.batch_bias, g.act = _p(bias), _p(batch_bias), act
    g.batch_ld = 0
    for t in (batch_bias, batch_scale):  # [B, N] views; may be column slices of one wide matrix
        if t is not None:
            if t.shape[0] != B or t.shape[1] != N or (g.batch_ld not in (0, _ld(t))):
                raise ValueError("conv_gemm: batch_bias / batch_scale must be [B, N] views sharing a row stride")
            g.batch_ld = _ld(t)
    g.res, g.ldr, g.res_dtype = _p(res), (0 if res is None else _ld(res)), (0 if res is None else _dt(res))
    g.res_scale, g.res_shift = _p(res_scale), _p(res_shift)
    g.out = _p(out)
    g.out_dtype, g.ldo = (0, 0) if out is None else (_dt(out), _ld(out))
    g.store_mode = store_mode
    g.blend_t, g.blend_f = _p(blend_t), _p(blend_f)
    g.ld_blend_in = 0 if blend_t is None else _ld(blend_t)
    if blend_f is not None and _ld(blend_f) != g.ld_blend_in:
        raise ValueError("conv_gemm: blend_t / blend_f must share a row stride")
    g.blend_out, g.ld_blend_out = _p(blend_out), (0 if blend_out is None else _ld(blend_out))
    g.compute = _dt(Wt)
    g.col_scale = _p(col_scale)
    g.batch_scale, g.ups = _p(batch_scale), ups
    if _SPLITK and -(-(B * g.OH * g.OW) // 128) * -(-N // 128) < 128 and KH == 1 and g.a_dtype == BF16:
        need = int(lib.uwir_conv_gemm_workspace_bytes(C.byref(g)))   # few output tiles: lend the split-K scratch
        if need:
            key = (A.device, _stream(A))
            ws = _GEMM_WS.get(key)
            if ws is None or ws.numel() * 4 < need:
                ws = torch.zeros(max(need // 4 + 1, 1 << 20), dtype=torch.float32, device=A.device)  # counters start at 0
                _GEMM_WS[key] = ws
            g.workspace, g.workspace_bytes = _p(ws), ws.numel() * 4
    _lib.check(lib.uwir_conv_gemm(C.byref(g), _stream(A)), "conv_gemm")


def fold_affine_weights(W: torch.Tensor, a: torch.Tensor, s: torch.Tensor, bias, Wb: torch.Tensor, bb: torch.Tensor) -> None:
    """Wb[b] = W * a[b] (columns), bb[b] = W s[b] + bias: the per-image LayerNorm2d affine folded into the next 1x1 conv."""
    _f32(W, a, s, bias, bb)
    N, K = W.shape
    Bn = a.shape[0]
    if (not W.is_contiguous() or tuple(a.shape) != (Bn, K) or tuple(s.shape) != (Bn, K) or not a.is_contiguous()
            or not s.is_contiguous() or Wb.dtype != torch.bfloat16 or tuple(Wb.shape) != (Bn, N, K) or not Wb.is_contiguous()
            or tuple(bb.shape) != (Bn, N) or not bb.is_contiguous()):
        raise ValueError("fold_affine_weights: W [N, K] fp32, a / s [B, K] fp32, Wb [B, N, K] bf16, bb [B, N] fp32, all contiguous")
    _lib.check(_lib.load().uwir_fold_affine_weights(_p(W), N, K, Bn, _p(a), _p(s), _p(bias), _p(Wb), _p(bb), _stream(W)),
               "fold_affine_weights")


def dwconv_ms(x: torch.Tensor, B: int, H: int, W: int, C_: int, taps, biases, out: torch.Tensor, in_scale=None,
              in_shift=None) -> None:
    lib = _lib.load()
    w3, w5, w7 = taps
    b3, b5, b7 = biases
    _f32(w3, w5, w7, b3, b5, b7, in_scale, in_shift)
    _lib.check(lib.uwir_dwconv_ms(_p(x), _dt(x), _ld(x), B, H, W, C_, _p(in_scale), _p(in_shift), _p(w3), _p(w5),
                                  _p(w7), _p(b3), _p(b5), _p(b7), _p(out), _dt(out), _ld(out), _stream(x)), "dwconv_ms")


_ATTN_WS = {}
_NO_FAST = False  # debugging aid (tools/): force the online-max attention loop


_PACKED = True  # tools/: False keeps the workspace at the header size, i.e. the LDS-staged kernels only


def _attn_workspace(dev, B: int, n: int, heads: int, hd: int, nv: int):
    """Scratch for the bf16 fast path per (device, stream), grown on demand: max_k |k|^2 per (b, head) + block marks,
    and for long hd 16 / 32 sequences the packed K / V^T streams (uwir.h).  Calls on one stream reuse it in order; a
    second stream (pipeline.ConditioningPrefetcher) gets its own."""
    lib = _lib.load()
    need = int(lib.uwir_flash_attn_packed_workspace_bytes(B, n, heads, hd, nv) if _PACKED
               else lib.uwir_flash_attn_workspace_bytes(B, n, heads))
    key = (dev, _stream(dev))
    t = _ATTN_WS.get(key)
    if t is None or t.numel() * 4 < need:
        t = torch.empty(max(need // 4, 4096), dtype=torch.float32, device=dev)
        _ATTN_WS[key] = t
    return t, need  # the library picks its kernels by the byte count it is GIVEN, not by the buffer's capacity


def flash_attn(q, k, v1, v2, B: int, n: int, heads: int, hd: int, scale: float, o1, o2, fast: bool = True) -> None:
    lib = _lib.load()
    dt = _dt(q)
    if _dt(k) != dt or _dt(v1) != dt or (v2 is not None and _dt(v2) != dt):
        raise TypeError("flash_attn: q, k, v must share a dtype")
    if o2 is not None and _dt(o2) != _dt(o1):
        raise TypeError("flash_attn: outputs must share a dtype")
    timed = _TIMING["on"]
    if timed:   # events around the call's MAIN kernel only, recorded by the library on the launch stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        ev1.record()   # (creates the underlying hipEvent_t handles)
        lib.uwir_flash_attn_probe(ev0.cuda_event, ev1.cuda_event)
    ws, ws_bytes = (_attn_workspace(q.device, B, n, heads, hd, 1 if v2 is None else 2)
                    if (fast and dt == BF16 and not _NO_FAST) else (None, 0))
    _lib.check(lib.uwir_flash_attn(_p(q), _ld(q), _p(k), _ld(k), _p(v1), _ld(v1), _p(v2),
                                   0 if v2 is None else _ld(v2), dt, B, n, heads, hd, scale, _p(o1), _ld(o1), _p(o2),
                                   0 if o2 is None else _ld(o2), _dt(o1), _p(ws), ws_bytes,
                                   _stream(q)), "flash_attn")
    if timed:
        _TIMING["events"].append((("flash_attn", B, n, heads, hd, 1 if v2 is None else 2, dt), ev0, ev1))


def fft_workspace_bytes(B: int, H: int, W: int, C_: int) -> int:
    return int(_lib.load().uwir_fft_workspace_bytes(B, H, W, C_))


def fft2_amp_phase(x, B: int, H: int, W: int, C_: int, amp, phase, workspace) -> None:
    _f32(x, amp, phase)
    lib = _lib.load()
    if _ld(amp) != _ld(phase):
        raise ValueError("fft2: amp / phase must share a row stride")
    _lib.check(lib.uwir_fft2_amp_phase(_p(x), _ld(x), B, H, W, C_, _p(amp), _p(phase), _ld(amp), _p(workspace),
                                       workspace.numel() * workspace.element_size(), _stream(x)), "fft2_amp_phase")


def ifft2_polar_add(amp, phase, addend, B: int, H: int, W: int, C_: int, out, workspace) -> None:
    _f32(amp, phase, addend, out)
    lib = _lib.load()
    if _ld(amp) != _ld(phase):
        raise ValueError("ifft2: amp / phase must share a row stride")
    _lib.check(lib.uwir_ifft2_polar_add(_p(amp), _p(phase), _ld(amp), _p(addend), _ld(addend), B, H, W, C_, _p(out),
                                        _ld(out), _p(workspace), workspace.numel() * workspace.element_size(),
                                        _stream(amp)), "ifft2_polar_add")


def bilinear_nchw_to_nhwc(src: torch.Tensor, dst: torch.Tensor, hd: int, wd: int) -> None:
    _f32(src, dst)
    lib = _lib.load()
    if not src.is_contiguous() or src.dim() != 4:
        raise ValueError("bilinear: src must be a contiguous NCHW tensor")
    B, M, hs, ws = src.shape
    _lib.check(lib.uwir_bilinear_nchw_to_nhwc(_p(src), B, M, hs, ws, _p(dst), _ld(dst), hd, wd, _stream(src)), "bilinear")


def bilinear_rows(src: torch.Tensor, B: int, hs: int, ws: int, C: int, dst: torch.Tensor, hd: int, wd: int) -> None:
    """F.interpolate(x, size=(hd, wd), mode="bilinear", align_corners=False) on NHWC rows: src [B*hs*ws, >= C] -> dst
    [B*hd*wd, >= C] of the same dtype (fp32 or bf16) -- the decoder's resize to its skip's size (S/model.py:150-151)."""
    if src.dtype != dst.dtype or src.shape[0] < B * hs * ws or dst.shape[0] < B * hd * wd:
        raise ValueError("bilinear_rows: src / dst dtype or row count mismatch")
    lib = _lib.load()
    _lib.check(lib.uwir_bilinear_rows(_p(src), _ld(src), B, hs, ws, C, _p(dst), _ld(dst), hd, wd, _dt(src), _stream(src)),
               "bilinear_rows")


def stem_conv3x3(x_nchw: torch.Tensor, w, bias, out: torch.Tensor) -> None:
    _f32(x_nchw, w, bias)
    lib = _lib.load()
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[-1]
    _lib.check(lib.uwir_stem_conv3x3(_p(x_nchw), B, Cin, H, W, _p(w), _p(bias), Cout, _p(out), _dt(out), _ld(out), _stream(x_nchw)),
               "stem_conv3x3")


def head_conv3x3(x: torch.Tensor, B: int, H: int, W: int, Cin: int, a_scale, a_shift, w, bias, Cout: int,
                 residual_nchw, out_nchw, preclamp_nchw=None) -> None:
    _f32(a_scale, a_shift, w, bias, residual_nchw, out_nchw, preclamp_nchw)
    lib = _lib.load()
    _lib.check(lib.uwir_head_conv3x3(_p(x), _dt(x), _ld(x), B, H, W, Cin, _p(a_scale), _p(a_shift), _p(w), _p(bias), Cout,
                                     _p(residual_nchw), _p(out_nchw), _p(preclamp_nchw), _stream(x)), "head_conv3x3")


def check_windows(rows, n_frames: int, FH: int, FW: int, T: int) -> None:
    """Host-side check of ``uwir_window`` records (a sequence of [frame, y, x, cy0, cy1, cx0, cx1, reserved] rows BEFORE
    they go to the device): the contract of include/uwir.h.  The kernels skip a window that breaks it; this raises."""
    for r in rows:
        f, y, x, cy0, cy1, cx0, cx1 = (int(v) for v in r[:7])
        if not (0 <= f < n_frames and 0 <= y <= FH - T and 0 <= x <= FW - T):
            raise ValueError(f"window {list(r)}: frame / origin outside {n_frames} frames of {FH} x {FW} at window edge {T}")
        if not (y <= cy0 <= cy1 <= y + T and x <= cx0 <= cx1 <= x + T):
            raise ValueError(f"window {list(r)}: owned core outside the window")


def _win_ok(win: torch.Tensor, frames: torch.Tensor, what: str) -> None:
    if (win.dtype != torch.int32 or win.dim() != 2 or win.shape[1] != 8 or not win.is_contiguous()
            or win.device != frames.device):
        raise ValueError(f"{what}: win must be a contiguous int32 [nwin, 8] tensor on the frames' device")


def stem_conv3x3_win(frames: torch.Tensor, win: torch.Tensor, T: int, w, bias, out: torch.Tensor) -> None:
    """Stem conv over the T x T windows `win` (int32 [nwin, 8] device descriptors, see uwir_window) of NCHW frames."""
    _f32(frames, w, bias)
    if frames.dim() != 4 or not frames.is_contiguous():
        raise ValueError("stem_conv3x3_win: frames must be contiguous NCHW")
    _win_ok(win, frames, "stem_conv3x3_win")
    NF, Cin, FH, FW = frames.shape
    _lib.check(_lib.load().uwir_stem_conv3x3_win(_p(frames), NF, Cin, FH, FW, _p(win), win.shape[0], T, _p(w), _p(bias),
                                                 w.shape[-1], _p(out), _dt(out), _ld(out), _stream(frames)), "stem_conv3x3_win")


def bilinear_win(frames: torch.Tensor, win: torch.Tensor, T: int, dst: torch.Tensor, hd: int, wd: int) -> None:
    _f32(frames, dst)
    if frames.dim() != 4 or not frames.is_contiguous():
        raise ValueError("bilinear_win: frames must be contiguous NCHW")
    _win_ok(win, frames, "bilinear_win")
    NF, M, FH, FW = frames.shape
    _lib.check(_lib.load().uwir_bilinear_win(_p(frames), NF, M, FH, FW, _p(win), win.shape[0], T, _p(dst), _ld(dst), hd, wd,
                                             _stream(frames)), "bilinear_win")


def head_conv3x3_win(x: torch.Tensor, T: int, Cin: int, a_scale, a_shift, w, bias, Cout: int, frames_in: torch.Tensor,
                     frames_out: torch.Tensor, win: torch.Tensor) -> None:
    _f32(a_scale, a_shift, w, bias, frames_in, frames_out)
    if (frames_in.dim() != 4 or frames_in.shape != frames_out.shape or not frames_in.is_contiguous()
            or not frames_out.is_contiguous()):
        raise ValueError("head_conv3x3_win: frames must be contiguous NCHW tensors of one shape")
    _win_ok(win, frames_in, "head_conv3x3_win")
    NF, _, FH, FW = frames_in.shape
    _lib.check(_lib.load().uwir_head_conv3x3_win(_p(x), _dt(x), _ld(x), win.shape[0], T, Cin, _p(a_scale), _p(a_shift), _p(w), _p(bias),
                                                 Cout, _p(frames_in), _p(frames_out), NF, FH, FW, _p(win), _stream(x)),
               "head_conv3x3_win")


def sde_euler_step(x, mu, noise, eps, x_out, theta: float, sigma: float, sigma_bar: float, dt: float,
                   score_weight: float = 1.0) -> None:
    _f32(x, mu, noise, eps, x_out)
    lib = _lib.load()
    _lib.check(lib.uwir_sde_euler_step(_p(x), _p(mu), _p(noise), _p(eps), _p(x_out), x.numel(), theta, sigma,
                                       sigma_bar, dt, score_weight, _stream(x)), "sde_euler_step")


def sde_posterior_step(x, mu, noise, eps, x_out, sigma_bar: float, expA: float, term1: float, term2: float,
                       std: float) -> None:
    _f32(x, mu, noise, eps, x_out)
    lib = _lib.load()
    _lib.check(lib.uwir_sde_posterior_step(_p(x), _p(mu), _p(noise), _p(eps), _p(x_out), x.numel(), sigma_bar, expA,
                                           term1, term2, std, _stream(x)), "sde_posterior_step")


def gemm(A: torch.Tensor, Wt: torch.Tensor, out: torch.Tensor, **kw) -> None:
    """Plain row-matrix GEMM out = epilogue(A @ Wt^T) through the conv entry point (1 x M "image")."""
    conv_gemm(A, Wt, out, B=1, H=1, W=A.shape[0], C_=Wt.shape[1], N=Wt.shape[0], **kw)


def ln_rows(x: torch.Tensor, gamma, beta, y: torch.Tensor, eps: float = 1e-5, add=None, add_period: int = 0,
            res=None) -> None:
    """y[m] = LayerNorm_C(x[m] + add[m % add_period]) * gamma + beta (+ res[m]); x, y fp32 / bf16 [M, C] views."""
    _f32(gamma, beta, add, res)
    lib = _lib.load()
    M, C_ = x.shape
    if y.shape[0] != M or y.shape[1] != C_ or gamma.numel() != C_:
        raise ValueError("ln_rows: shape mismatch")
    if add is not None and (add.shape[0] != add_period or add.shape[1] != C_):
        raise ValueError("ln_rows: add table must be [add_period, C]")
    _lib.check(lib.uwir_ln_rows(_p(x), _dt(x), _ld(x), M, C_, _p(add), add_period, 0 if add is None else _ld(add), _p(gamma),
                                _p(beta), eps, _p(res), 0 if res is None else _ld(res), _p(y), _dt(y), _ld(y),
                                _stream(x)), "ln_rows")


def patchify(img: torch.Tensor, P: int, out: torch.Tensor) -> None:
    _f32(img)
    lib = _lib.load()
    if img.dim() != 4 or not img.is_contiguous():
        raise ValueError("patchify: image must be a contiguous NCHW tensor")
    B, Cin, H, W = img.shape
    _lib.check(lib.uwir_patchify(_p(img), B, Cin, H, W, P, _p(out), _dt(out), _ld(out), _stream(img)), "patchify")


def add_rows(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> None:
    """out = a + b; a fp32, b and out fp32 or bf16."""
    _f32(a)
    lib = _lib.load()
    M, C_ = a.shape
    if b.shape != a.shape or out.shape != a.shape:
        raise ValueError("add_rows: shape mismatch")
    _lib.check(lib.uwir_add_rows(_p(a), _ld(a), _p(b), _dt(b), _ld(b), _p(out), _dt(out), _ld(out), M, C_, _stream(a)),
               "add_rows")


def unet_input(xt: torch.Tensor, cond: torch.Tensor, Hp: int, Wp: int, out: torch.Tensor) -> None:
    """out [B*Hp*Wp, Cp] = NHWC cat([xt - cond, cond]) with reflect padding to (Hp, Wp), channels zero-filled to Cp."""
    _f32(xt, cond, out)
    lib = _lib.load()
    if xt.shape != cond.shape or xt.dim() != 4 or not xt.is_contiguous() or not cond.is_contiguous():
        raise ValueError("unet_input: xt / cond must be contiguous NCHW tensors of one shape")
    B, Cin, H, W = xt.shape
    if not out.is_contiguous() or out.shape[0] != B * Hp * Wp:
        raise ValueError("unet_input: bad output buffer")
    _lib.check(lib.uwir_unet_input(_p(xt), _p(cond), B, Cin, H, W, Hp, Wp, out.shape[1], _p(out), _stream(xt)), "unet_input")


def groupnorm_affine(x: torch.Tensor, B: int, n: int, C_: int, groups: int, eps: float, gamma, beta, a_out, s_out,
                     workspace: torch.Tensor) -> None:
    _f32(x, gamma, beta, a_out, s_out)
    lib = _lib.load()
    _lib.check(lib.uwir_groupnorm_affine(_p(x), _ld(x), B, n, C_, groups, eps, _p(gamma), _p(beta), _p(a_out), _p(s_out),
                                         _p(workspace), workspace.numel() * workspace.element_size(), _stream(x)),
               "groupnorm_affine")


def linattn_workspace_bytes(B: int, n: int, heads: int) -> int:
    return int(_lib.load().uwir_linattn_workspace_bytes(B, n, heads))


def linear_attention(qkv: torch.Tensor, B: int, n: int, heads: int, out: torch.Tensor, workspace: torch.Tensor) -> None:
    lib = _lib.load()
    _lib.check(lib.uwir_linear_attention(_p(qkv), _dt(qkv), _ld(qkv), B, n, heads, _p(out), _dt(out), _ld(out),
                                         _p(workspace), workspace.numel() * workspace.element_size(), _stream(qkv)),
               "linear_attention")


def linattn_block_workspace_bytes(B: int, n: int) -> int:
    return int(_lib.load().uwir_linattn_block_workspace_bytes(B, n))


def linattn_block(x: torch.Tensor, B: int, n: int, dim: int, g_pre, w_qkv, w_out, b_out, g_out, out: torch.Tensor,
                  workspace: torch.Tensor, eps: float = 1e-5, heads: int = 4) -> None:
    """out = LN_gout(W_out . LinearAttention(W_qkv . LN_gpre(x)) + b_out) + x, fused (bf16 MFMA); see uwir.h."""
    _f32(g_pre, b_out, g_out)
    if _dt(x) != _dt(out):
        raise TypeError("linattn_block: x and out must share a dtype (fp32 or bf16)")
    if w_qkv.dtype != torch.bfloat16 or w_out.dtype != torch.bfloat16 or not w_qkv.is_contiguous() or not w_out.is_contiguous():
        raise TypeError("linattn_block: weights must be contiguous bf16")
    if tuple(w_qkv.shape) != (3 * heads * 32, dim) or tuple(w_out.shape) != (dim, heads * 32):
        raise ValueError("linattn_block: weight shapes must be [3*heads*32, dim] and [dim, heads*32]")
    lib = _lib.load()
    _lib.check(lib.uwir_linattn_block(_p(x), _dt(x), _ld(x), B, n, dim, heads, _p(g_pre), _p(w_qkv), _p(w_out), _p(b_out), _p(g_out),
                                      eps, _p(out), _ld(out), _p(workspace), workspace.numel() * workspace.element_size(),
                                      _stream(x)), "linattn_block")


def geglu(h: torch.Tensor, out: torch.Tensor) -> None:
    lib = _lib.load()
    M, F2 = h.shape
    if _dt(h) != _dt(out) or out.shape[0] != M or out.shape[1] * 2 != F2:
        raise ValueError("geglu: h must be [M, 2F] and out [M, F] of one dtype")
    _lib.check(lib.uwir_geglu(_p(h), _dt(h), _ld(h), M, F2 // 2, _p(out), _ld(out), _stream(h)), "geglu")


def softmax_rows(x: torch.Tensor, mul, y: torch.Tensor) -> None:
    _f32(x, mul, y)
    lib = _lib.load()
    B, C_ = x.shape
    _lib.check(lib.uwir_softmax_rows(_p(x), _ld(x), B, C_, _p(mul), _p(y), _ld(y), _stream(x)), "softmax_rows")


def tail_conv3x3(x: torch.Tensor, B: int, H: int, W: int, Cin: int, w, bias, Cout: int, out_nchw: torch.Tensor,
                 split: bool = False) -> None:
    """``split`` (fp32 rows only): bf16 x 3 products on the matrix cores instead of exact fp32 (uwir_tail_conv3x3_split)."""
    _f32(w, bias, out_nchw)
    lib = _lib.load()
    Ho, Wo = out_nchw.shape[2], out_nchw.shape[3]
    if split and x.dtype == torch.float32:
        _lib.check(lib.uwir_tail_conv3x3_split(_p(x), _ld(x), B, H, W, Cin, _p(w), _p(bias), Cout, _p(out_nchw), Ho, Wo,
                                               _stream(x)), "tail_conv3x3_split")
        return
    _lib.check(lib.uwir_tail_conv3x3(_p(x), _dt(x), _ld(x), B, H, W, Cin, _p(w), _p(bias), Cout, _p(out_nchw), Ho, Wo, _stream(x)),
               "tail_conv3x3")


def affine_cast(x: torch.Tensor, B: int, n: int, C_: int, a, s, out: torch.Tensor) -> None:
    """out[m, c] = a[b, c] * x[m, c] + s[b, c] (b = m // n); x and out fp32 / bf16."""
    _f32(a, s)
    lib = _lib.load()
    _lib.check(lib.uwir_affine_cast(_p(x), _dt(x), _ld(x), B, n, C_, _p(a), _p(s), _p(out), _dt(out), _ld(out), _stream(x)),
               "affine_cast")


def image_metrics(x: torch.Tensor, y: torch.Tensor, c1: float = 0.01 ** 2, c2: float = 0.03 ** 2) -> torch.Tensor:
    """-> [B, 3] = per image (ssim_loss, mse, psnr) of NCHW fp32 batches; see uwir_image_metrics."""
    _f32(x, y)
    lib = _lib.load()
    if x.shape != y.shape or x.dim() != 4 or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("image_metrics: x and y must be contiguous NCHW tensors of one shape")
    B, C_, H, W = x.shape
    ws = torch.empty(int(lib.uwir_image_metrics_workspace_bytes(B, C_, H, W)) // 8, dtype=torch.float64, device=x.device)
    out = torch.empty(B, 3, dtype=torch.float32, device=x.device)
    _lib.check(lib.uwir_image_metrics(_p(x), _p(y), B, C_, H, W, c1, c2, _p(out), _p(ws), ws.numel() * 8, _stream(x)),
               "image_metrics")
    return out


def _census_wrap(name: str, fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        if not _TIMING["all"]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        _TIMING["ops"].append((name, a, k, e0, e1))
        return r
    return wrapper


for _n in ("dwt2", "idwt2", "ln2d_affine", "conv_gemm", "dwconv_ms", "flash_attn", "fft2_amp_phase", "ifft2_polar_add",
           "affine_cast", "stem_conv3x3", "head_conv3x3", "sde_euler_step", "sde_posterior_step", "linear_attention",
           "ln_rows", "groupnorm_affine", "geglu", "linattn_block", "tail_conv3x3", "unet_input", "bilinear_nchw_to_nhwc", "linear",
           "gate_idwt", "gate_blend", "ffn_fused", "fold_affine_weights"):
    globals()[_n] = _census_wrap(_n, globals()[_n])
del _n
