"""The WTB qkv / ffn.0 GEMM shapes with plain weights + bias against per-image weight panels + per-image bias (the folded
LayerNorm2d affine), A/B/A/B.   python tools/exp/qkv_gemm.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
bf = torch.bfloat16
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
for _ in range(100): a @ b
B = 16
for (hw, K, N) in ((128, 192, 576), (64, 384, 1152), (32, 768, 2304), (128, 192, 384), (64, 384, 768)):
    M = B * hw * hw
    x = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    wb = (torch.randn(B, N, K, device="cuda") / K ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda"); bb = torch.randn(B, N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=bf)
    fa = lambda: ops.conv_gemm(x, w, out, B=B, H=hw, W=hw, C_=K, N=N, bias=bias)
    fb = lambda: ops.conv_gemm(x, wb, out, B=B, H=hw, W=hw, C_=K, N=N, batch_bias=bb, w_batched=True)
    fc = lambda: ops.conv_gemm(x, w, out, B=B, H=hw, W=hw, C_=K, N=N, batch_bias=bb)
    ta, tb, tc, ta2, tb2 = timeit(fa), timeit(fb), timeit(fc), timeit(fa), timeit(fb)
    gb = (M * K * 2 + M * N * 2) / 1e3
    print(f"M{M} N{N} K{K}: plain {ta:.1f}/{ta2:.1f} us ({gb / ta2:.0f} GB/s)  batched {tb:.1f}/{tb2:.1f} us ({gb / tb2:.0f} GB/s)  plain+batch_bias {tc:.1f}")
