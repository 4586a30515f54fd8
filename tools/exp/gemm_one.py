"""A few launches of the output-heavy 1x1 GEMMs of level 0 (for PMC passes): python tools/exp/gemm_one.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops, _lib
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
variants = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
M = 262144
g = torch.Generator().manual_seed(0)
for (K, N, res) in ((192, 576, False), (384, 192, True), (192, 384, False)):
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, generator=g).cuda().bfloat16() if res else None
    bias = torch.randn(N, generator=g).cuda()
    for rnd in range(2 if len(variants) > 1 else 1):
      for var in variants:
        lib.uwir_conv_variant(var)
        for _ in range(2):
            ops.conv_gemm(A, W, out, B=16, H=128, W=128, C_=K, N=N, bias=bias, res=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv_gemm(A, W, out, B=16, H=128, W=128, C_=K, N=N, bias=bias, res=r)
        e1.record()
        torch.cuda.synchronize()
        by = M * (K + N * (2 if res else 1)) * 2
        print(f"variant {var} K{K} N{N} res={res}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us, algorithmic {by / 1e6:.0f} MB")
    lib.uwir_conv_variant(0)
