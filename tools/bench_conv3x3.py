"""A/B of the 3x3 halo-tile convolution kernel at the IR-SDE UNet's fine-level shapes (B = 8): 8-row tiles (variant 0) against
16-row tiles where eligible (variant 1), interleaved rounds in one process, results compared.
    python tools/bench_conv3x3.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import _lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
B = 8
for (S, C, N, adt, odt, res) in ((256, 64, 64, torch.float32, torch.bfloat16, False), (256, 64, 64, torch.bfloat16, torch.float32, True),
                                 (128, 64, 64, torch.float32, torch.bfloat16, False), (256, 128, 64, torch.float32, torch.bfloat16, False),
                                 (128, 128, 128, torch.bfloat16, torch.float32, True), (64, 128, 128, torch.float32, torch.bfloat16, False)):
    g = torch.Generator().manual_seed(0)
    M = B * S * S
    x = torch.randn(M, C, generator=g).cuda().to(adt)
    w = (torch.randn(N, 9 * C, generator=g) / (9 * C) ** 0.5).cuda().bfloat16()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    ss = torch.randn(B, 2 * N, generator=g).cuda()
    outs, times = {}, {}
    variants = (0, 1)
    for v in variants:
        lib.uwir_conv_variant(v)
        outs[v] = torch.zeros(M, N, device="cuda", dtype=odt)
        times[v] = []
        ops.conv_gemm(x, w, outs[v], B=B, H=S, W=S, C_=C, N=N, KH=3, KW=3, pad=1, act=ops.ACT_SILU, res=r,
                      batch_scale=None if res else ss[:, :N], batch_bias=None if res else ss[:, N:])
    torch.cuda.synchronize()
    diff = float((outs[variants[0]].float() - outs[variants[1]].float()).abs().max())
    for rnd in range(3):
        for v in variants:
            lib.uwir_conv_variant(v)
            f = lambda: ops.conv_gemm(x, w, outs[v], B=B, H=S, W=S, C_=C, N=N, KH=3, KW=3, pad=1, act=ops.ACT_SILU, res=r,  # noqa: E731
                                      batch_scale=None if res else ss[:, :N], batch_bias=None if res else ss[:, N:])
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / reps)
    lib.uwir_conv_variant(1)
    by = x.numel() * x.element_size() + M * N * outs[variants[0]].element_size() + (M * N * 4 if res else 0)
    for v in variants:
        ms = min(times[v])
        print(f"conv3x3 {S}x{S} C{C}->N{N} {str(adt)[6:]}->{str(odt)[6:]} res={res} variant {v}: {ms * 1e3:7.1f} us  "
              f"{2.0 * M * N * 9 * C / ms / 1e9:6.0f} TF/s  {by / ms / 1e6:5.0f} GB/s", flush=True)
    print(f"   max |diff| between variants: {diff:.3e}", flush=True)
