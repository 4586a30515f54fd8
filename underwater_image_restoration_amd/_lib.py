"""ctypes binding of libuwir_hip.so (the C ABI declared in include/uwir.h).

The product path has NO fallback: if the shared library is missing or a call
fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuwir_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_SILU = 0, 1, 2, 3, 4

vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class GemmArgs(C.Structure):
    """Mirror of ``struct uwir_gemm_args`` (include/uwir.h) -- field order is ABI."""
    _fields_ = [
        ("A", vp), ("a_dtype", ci), ("lda", ci),
        ("B", ci), ("H", ci), ("W", ci), ("C", ci),
        ("KH", ci), ("KW", ci), ("stride", ci), ("pad", ci),
        ("OH", ci), ("OW", ci),
        ("a_scale", vp), ("a_shift", vp),
        ("Wt", vp), ("w_dtype", ci), ("N", ci),
        ("bias", vp), ("batch_bias", vp), ("act", ci),
        ("res", vp), ("ldr", ci),
        ("res_scale", vp), ("res_shift", vp),
        ("out", vp), ("out_dtype", ci), ("ldo", ci),
        ("store_mode", ci),
        ("blend_t", vp), ("blend_f", vp), ("ld_blend_in", ci),
        ("blend_out", vp), ("ld_blend_out", ci),
        ("compute", ci),
        ("col_scale", vp),
        ("batch_scale", vp),
        ("ups", ci),
        ("batch_ld", ci),
        ("workspace", vp), ("workspace_bytes", sz),
        ("res_dtype", ci),
        ("w_batch_stride", C.c_longlong),
    ]


class TeacherArgs(C.Structure):
    """Mirror of ``struct uwir_teacher_args`` (include/uwir.h) -- field order is ABI."""
    _fields_ = [
        ("tokens", vp), ("ld_tokens", ci), ("text", vp), ("thresholds", vp), ("image", vp),
        ("B", ci), ("h", ci), ("w", ci), ("C", ci), ("K", ci),
        ("img_c", ci), ("img_h", ci), ("img_w", ci),
        ("temperature", cf), ("alpha", cf), ("confidence_delta", cf), ("eps", cf),
        ("morphology_kernel", ci),
        ("use_crf", ci), ("crf_iters", ci), ("crf_ks", ci),
        ("crf_sigma", cf), ("crf_edge_weight", cf),
        ("scores", vp), ("probs", vp), ("masks", vp), ("intensity", vp), ("raw_probs", vp),
        ("avg_intensity", vp), ("global_prob", vp), ("confidence", vp),
        ("logits", vp), ("code", vp), ("confidence_scale", vp),
        ("workspace", vp), ("workspace_bytes", sz),
    ]


_SIGS = {
    "uwir_version": (ci, []),
    "uwir_last_error": (C.c_char_p, []),
    "uwir_dwt2": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp]),
    "uwir_idwt2": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, ci, ci, vp]),
    "uwir_ln2d_workspace_bytes": (sz, [ci, ci, ci]),
    "uwir_ln2d_affine": (ci, [vp, ci, ci, ci, ci, ci, cf, vp, vp, vp, ci, vp, vp, vp, vp, vp, sz, vp]),
    "uwir_ln2d_noalloc_max_bytes": (sz, []),
    "uwir_ln2d_affine_noalloc": (ci, [vp, ci, ci, ci, ci, ci, cf, vp, vp, vp, ci, vp, vp, vp, vp, vp]),
    "uwir_affine_cast": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, ci, ci, vp]),
    "uwir_chan_attn_fold": (ci, [vp, vp, vp, vp, vp, ci, vp]),
    "uwir_linear": (ci, [vp, ci, ci, ci, vp, vp, ci, ci, vp, ci, vp]),
    "uwir_conv_gemm_workspace_bytes": (sz, [C.POINTER(GemmArgs)]),
    "uwir_conv_gemm": (ci, [C.POINTER(GemmArgs), vp]),
    "uwir_dwconv_ms": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    "uwir_flash_attn_probe": (ci, [vp, vp]),
    "uwir_flash_attn_workspace_bytes": (sz, [ci, ci, ci]),
    "uwir_flash_attn_packed_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "uwir_flash_attn": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, cf, vp, ci, vp, ci, ci, vp, sz, vp]),
    "uwir_fft_workspace_bytes": (sz, [ci, ci, ci, ci]),
    "uwir_fft2_amp_phase": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, sz, vp]),
    "uwir_ifft2_polar_add": (ci, [vp, vp, ci, vp, ci, ci, ci, ci, ci, vp, ci, vp, sz, vp]),
    "uwir_bilinear_nchw_to_nhwc": (ci, [vp, ci, ci, ci, ci, vp, ci, ci, ci, vp]),
    "uwir_bilinear_rows": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, vp]),
    "uwir_stem_conv3x3": (ci, [vp, ci, ci, ci, ci, vp, vp, ci, vp, ci, ci, vp]),
    "uwir_head_conv3x3": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
    "uwir_stem_conv3x3_win": (ci, [vp, ci, ci, ci, ci, vp, ci, ci, vp, vp, ci, vp, ci, ci, vp]),
    "uwir_bilinear_win": (ci, [vp, ci, ci, ci, ci, vp, ci, ci, vp, ci, ci, ci, vp]),
    "uwir_head_conv3x3_win": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, ci, ci, vp, vp]),
    "uwir_sde_euler_step": (ci, [vp, vp, vp, vp, vp, sz, cf, cf, cf, cf, cf, vp]),
    "uwir_sde_posterior_step": (ci, [vp, vp, vp, vp, vp, sz, cf, cf, cf, cf, cf, vp]),
    "uwir_ln_rows": (ci, [vp, ci, sz, ci, ci, vp, ci, ci, vp, vp, cf, vp, sz, vp, ci, sz, vp]),
    "uwir_unet_input": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "uwir_groupnorm_affine": (ci, [vp, ci, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, sz, vp]),
    "uwir_linattn_workspace_bytes": (sz, [ci, ci, ci]),
    "uwir_linear_attention": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, ci, vp, sz, vp]),
    "uwir_linattn_block_workspace_bytes": (sz, [ci, ci]),
    "uwir_linattn_block": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, cf, vp, ci, vp, sz, vp]),
    "uwir_conv_variant": (None, [ci]),
    "uwir_dwconv_tiles_per_block": (None, [ci]),
    "uwir_attn_variant": (None, [ci]),
    "uwir_geglu": (ci, [vp, ci, sz, ci, ci, vp, sz, vp]),
    "uwir_softmax_rows": (ci, [vp, ci, ci, ci, vp, vp, ci, vp]),
    "uwir_tail_conv3x3": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, ci, ci, vp]),
    "uwir_head_variant": (None, [ci]),
    "uwir_gemm_wide_probe": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "uwir_gate_idwt_supported": (ci, [ci, ci]),
    "uwir_ffn_fused_supported": (ci, [ci, ci]),
    "uwir_ffn_fused": (ci, [vp, ci, vp, C.c_longlong, vp, vp, vp, ci, ci, ci, vp, ci, vp]),
    "uwir_gate_idwt": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, vp, vp]),
    "uwir_gate_blend": (ci, [vp, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, vp]),
    "uwir_fold_affine_weights": (ci, [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
    "uwir_tail_conv3x3_split": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, ci, ci, vp]),
    "uwir_image_metrics_workspace_bytes": (sz, [ci, ci, ci, ci]),
    "uwir_image_metrics": (ci, [vp, vp, ci, ci, ci, ci, cf, cf, vp, vp, sz, vp]),
    "uwir_tensor2img_u8": (ci, [vp, ci, ci, ci, cf, cf, ci, vp, vp]),
    "uwir_sqdiff_u8": (ci, [vp, vp, sz, vp, vp]),
    "uwir_teacher_heads_workspace_bytes": (sz, [ci, ci, ci]),
    "uwir_teacher_heads": (ci, [C.POINTER(TeacherArgs), vp]),
    "uwir_guided_crf": (ci, [C.POINTER(TeacherArgs), vp, vp, vp]),
    "uwir_cv_workspace_bytes": (sz, [ci, ci, ci, ci]),
    "uwir_histogram_equalization": (ci, [vp, ci, ci, ci, vp, vp, sz, vp]),
    "uwir_clahe_equalization": (ci, [vp, ci, ci, cf, ci, ci, vp, vp, sz, vp]),
    "uwir_cv_scores": (ci, [vp, ci, ci, vp, vp, sz, vp]),
    "uwir_gamma_correction": (ci, [vp, vp, sz, cf, vp]),
    "uwir_lf_hf_mix": (ci, [vp, vp, C.c_double, vp, sz, vp]),
    "uwir_wavelet_denoise_workspace_bytes": (sz, [ci, ci, ci]),
    "uwir_wavelet_denoise": (ci, [vp, ci, ci, ci, cf, vp, vp, sz, vp]),
    "uwir_guided_filter_workspace_bytes": (sz, [ci, ci]),
    "uwir_guided_filter": (ci, [vp, vp, ci, ci, ci, cf, vp, vp, sz, vp]),
    "uwir_multi_scale_retinex_workspace_bytes": (sz, [ci, ci, ci]),
    "uwir_multi_scale_retinex": (ci, [vp, ci, ci, ci, C.POINTER(C.c_double), ci, vp, vp, sz, vp]),
    "uwir_enhance_state_workspace_bytes": (sz, []),
    "uwir_enhance_state": (ci, [vp, ci, ci, ci, vp, vp, sz, vp]),
    "uwir_add_rows": (ci, [vp, sz, vp, ci, sz, vp, ci, sz, ci, ci, vp]),
    "uwir_patchify": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, sz, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


class UwirError(RuntimeError):
    pass


def load():
    """dlopen the HIP library once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UwirError(f"{LIB_PATH} is missing: build it with `python -m underwater_image_restoration_amd.build` "
                        "(hipcc --offload-arch=gfx950). There is no fallback path.")
    # torch ships its own libamdhip64.so; import it FIRST so this library binds to the same HIP runtime
    # instance (two runtimes in one process => "no ROCm-capable device is detected" in the second).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().uwir_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise UwirError(f"{what}: rc={rc}: {msg}")
