"""Per-launch timing table of one forward (HIP events around every ops.* call, tagged with shapes).

    python tools/profile_ops.py [--workload student|unet] [--precision bf16|fp32] [--batch 16] [--size 256]
Prints launches aggregated by (op, shape) sorted by total time, with achieved TFLOP/s for GEMM / attention.
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.synth import (student_param_shapes, synth_state_dict, synth_student_inputs,  # noqa: E402
                                                    unet_param_shapes)

EVENTS = []


def wrap(name, tagger):
    orig = getattr(ops, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        EVENTS.append((name, tagger(*a, **k), e0, e1))
        return r
    setattr(ops, name, f)


def gemm_tag(A, Wt, out, **k):
    M = k["B"] * (k.get("OH") or ((k["H"] << k.get("ups", 0)) + 2 * k.get("pad", 0) - k.get("KH", 1)) // k.get("stride", 1) + 1) * \
        (k.get("OW") or ((k["W"] << k.get("ups", 0)) + 2 * k.get("pad", 0) - k.get("KW", 1)) // k.get("stride", 1) + 1)
    K = Wt.shape[1]
    N = k["N"]
    extra = ("a=" + str(A.dtype).split(".")[-1][:4] + ("+aff" if k.get("a_scale") is not None else "")
             + f" k{k.get('KH', 1)}" + (" o=" + str(out.dtype).split(".")[-1][:4] if out is not None else "")
             + "".join(" " + nm for nm in ("bias", "batch_bias", "batch_scale", "res", "res_scale", "col_scale", "blend_out")
                       if k.get(nm) is not None)
             + (f" act{k['act']}" if k.get("act") else "") + (" d2s" if k.get("store_mode") else ""))
    return (f"M{M} N{N} K{K} {extra}", 2.0 * M * N * K, 0)


def attn_tag(q, k, v1, v2, B, n, heads, hd, scale, o1, o2):
    nv = 1 if v2 is None else 2
    return (f"B{B} n{n} h{heads} hd{hd} nv{nv}", (2 + 2 * nv) * B * heads * float(n) * n * hd, 0)


def generic(name):
    def t(*a, **k):
        shp = [tuple(x.shape) for x in a if torch.is_tensor(x)][:2]
        by = sum(x.numel() * x.element_size() for x in list(a) + list(k.values()) if torch.is_tensor(x))
        return (str(shp), 0.0, by)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="student")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-fast-attn", action="store_true")
    args = ap.parse_args()
    ops._NO_FAST = args.no_fast_attn
    wrap("conv_gemm", gemm_tag)
    wrap("flash_attn", attn_tag)
    for nm in ("dwt2", "idwt2", "ln2d_affine", "dwconv_ms", "fft2_amp_phase", "ifft2_polar_add", "stem_conv3x3", "head_conv3x3",
               "linear", "bilinear_nchw_to_nhwc", "chan_attn_fold", "ln_rows", "linear_attention", "groupnorm_affine", "geglu",
               "unet_input", "tail_conv3x3", "add_rows", "softmax_rows", "patchify"):
        wrap(nm, generic(nm))
    if args.workload == "student":
        from underwater_image_restoration_amd.student import NAFNetWFIGate
        mk = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
        model = NAFNetWFIGate(**mk, precision=args.precision)
        model.load_state_dict(synth_state_dict(student_param_shapes(**mk), seed=0))
        model = model.cuda().eval()
        x, z, m = (t.cuda() for t in synth_student_inputs(args.batch, args.size, args.size, 512, 10, mask_hw=(7, 7)))
        run = lambda: model(x, z, m)  # noqa: E731
    elif args.workload == "daclip":
        from underwater_image_restoration_amd.daclip import DaCLIPVision
        from underwater_image_restoration_amd.synth import daclip_vision_param_shapes
        clip = DaCLIPVision(precision=args.precision)
        clip.load_state_dict(synth_state_dict(daclip_vision_param_shapes(), seed=1))
        clip = clip.cuda().eval()
        cin = torch.randn(args.batch, 3, 224, 224, device="cuda")
        run = lambda: clip.encode_image(cin, control=True)  # noqa: E731
    else:
        from underwater_image_restoration_amd.unet import ConditionalUNet
        model = ConditionalUNet(3, 3, 64, [1, 2, 4, 8], 512, True, True, precision=args.precision)
        model.load_state_dict(synth_state_dict(unet_param_shapes(), seed=2))
        model = model.cuda().eval()
        B = min(args.batch, 8)
        x = torch.rand(B, 3, args.size, args.size, device="cuda")
        c = torch.randn(B, 512, device="cuda")
        run = lambda: model(x + 0.1, x, 50, text_context=c, image_context=c)  # noqa: E731
    run()
    run()
    torch.cuda.synchronize()
    EVENTS.clear()
    run()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    total = 0.0
    for name, (tag, fl, by), e0, e1 in EVENTS:
        ms = e0.elapsed_time(e1)
        total += ms
        a = agg.setdefault((name, tag), [0, 0.0, fl, by])
        a[0] += 1
        a[1] += ms
    print(f"total {total:.2f} ms over {len(EVENTS)} launches ({args.workload} {args.precision} B{args.batch} {args.size}^2)")
    for (name, tag), (cnt, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        rate = f"{fl * cnt / ms / 1e9:8.0f} TF/s" if fl else (f"{by * cnt / ms / 1e6:8.0f} GB/s*" if by else "")
        print(f"{ms:8.3f} ms {100 * ms / total:5.1f}%  x{cnt:<3d} {ms / cnt:7.3f}  {name:16s} {tag:84s} {rate}")


if __name__ == "__main__":
    main()
