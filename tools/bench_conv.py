"""Micro-benchmark of uwir_conv_gemm on the IR-SDE UNet's convolution shapes (B = 8, 256 x 256, nf = 64, [1, 2, 4, 8]),
timed A/B/A/B over uwir_conv_variant values:  python tools/bench_conv.py [reps] [variants, e.g. 0,8] [shape indices, e.g. 0,15]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import _lib, ops  # noqa: E402

f32, bf = torch.float32, torch.bfloat16
# (H, C, N, k, a dtype, out dtype, residual, ups)
SHAPES = (
    (256, 64, 64, 3, bf, bf, False, 0), (256, 64, 64, 3, bf, bf, True, 0), (256, 128, 64, 3, bf, bf, False, 0),
    (256, 128, 64, 1, bf, bf, False, 0),
    (128, 64, 64, 3, bf, bf, False, 0), (128, 64, 64, 3, bf, bf, True, 0), (128, 192, 128, 3, bf, bf, False, 0),
    (128, 128, 128, 3, bf, bf, True, 0),
    (64, 128, 128, 3, bf, bf, False, 0), (64, 128, 128, 3, bf, bf, True, 0), (64, 384, 256, 3, bf, bf, False, 0),
    (64, 256, 256, 3, bf, bf, True, 0),
    (32, 256, 256, 3, bf, bf, False, 0), (32, 256, 256, 3, bf, bf, True, 0), (32, 512, 512, 3, bf, bf, False, 0),
    (32, 512, 512, 3, bf, bf, True, 0), (32, 768, 512, 3, bf, bf, False, 0), (32, 512, 256, 3, bf, bf, False, 1),
    (64, 256, 128, 3, bf, bf, False, 1), (128, 128, 64, 3, bf, bf, False, 1),
)


def timeit(f, reps):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    if os.environ.get("UWIR_EXP_LIB"):      # an experimental build of the library (ablations under tools/exp)
        _lib.LIB_PATH = os.path.abspath(os.environ["UWIR_EXP_LIB"])
    lib = _lib.load()
    B = 8
    shapes = [SHAPES[int(i)] for i in sys.argv[3].split(",")] if len(sys.argv) > 3 else SHAPES
    a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
    for _ in range(200):   # warm the chip up: the first thing timed in a process reads 10-15 % slow
        a @ b
    for H, C, N, k, adt, odt, res, ups in shapes:
        OH = H << ups
        x = torch.randn(B * H * H, C, device="cuda").to(adt)
        w = (torch.randn(N, k * k * C, device="cuda") / (k * k * C) ** 0.5).to(bf)
        out = torch.empty(B * OH * OH, N, device="cuda", dtype=odt)
        r = torch.randn(B * OH * OH, N, device="cuda").to(bf) if res else None
        ss = torch.randn(B, 2 * N, device="cuda")
        kw = dict(B=B, H=H, W=H, C_=C, N=N, KH=k, KW=k, pad=k // 2, ups=ups)
        if k == 3 and not res and not ups:
            kw.update(batch_scale=ss[:, :N], batch_bias=ss[:, N:], act=ops.ACT_SILU)
        elif res:
            kw.update(act=ops.ACT_SILU, res=r)
        f = lambda: ops.conv_gemm(x, w, out, **kw)  # noqa: E731
        outs, times = {}, {v: [] for v in variants}
        for rnd in range(2):
            for v in variants:
                lib.uwir_conv_variant(v)
                if rnd == 0:
                    out.zero_()
                    f()
                    outs[v] = out.float().clone()
                times[v].append(timeit(f, reps))
        lib.uwir_conv_variant(0)
        fl = 2.0 * B * OH * OH * N * k * k * C
        by = x.numel() * x.element_size() + out.numel() * out.element_size() + (r.numel() * 2 if res else 0)
        txt = "  ".join(f"v{v}: " + "/".join(f"{t:.1f}" for t in times[v]) + f" us ({fl / min(times[v]) / 1e6:.0f} TF/s)" for v in variants)
        dif = max(float((outs[v] - outs[variants[0]]).abs().max()) for v in variants)
        print(f"{H:3d}^2 C{C:<3d}->N{N:<3d} k{k} a={str(adt)[6:]:8s} out={str(odt)[6:]:8s} res={int(res)} ups={ups}: {txt}  "
              f"{by / min(times[variants[0]]) / 1e3:.0f} GB/s  max diff {dif:.2e}", flush=True)


if __name__ == "__main__":
    main()
