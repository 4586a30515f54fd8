"""Micro-benchmark of uwir_conv_gemm on 1x1 shapes: python tools/bench_gemm.py M N K [act] [out=bf16|f32] [a=bf16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402


def run(M, N, K, act=0, odt=torch.bfloat16, adt=torch.bfloat16, res=False, reps=20, wdt=torch.bfloat16):
    A = torch.randn(M, K, device="cuda").to(adt)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(wdt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=odt)
    r = torch.randn(M, N, device="cuda") if res else None
    f = lambda: ops.conv_gemm(A, W, out, B=1, H=1, W=M, C_=K, N=N, bias=bias, act=act, res=r)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = M * K * A.element_size() + M * N * out.element_size() + (M * N * 4 if res else 0)
    print(f"M{M} N{N} K{K} act{act} out={str(odt)[6:]} a={str(adt)[6:]} w={str(wdt)[6:]} res={res}: {ms:.3f} ms  "
          f"{2.0 * M * N * K / ms / 1e9:7.0f} TF/s  {by / ms / 1e6:6.0f} GB/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        from underwater_image_restoration_amd import _lib
        lib = _lib.load()
        for M, N, K in ((262144, 576, 1728), (65536, 1152, 3456), (16384, 2304, 6912), (262144 - 100, 192, 1024)):
            A = torch.randn(M, K, device="cuda").bfloat16()
            W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
            bias, cs = torch.randn(N, device="cuda"), torch.rand(N, device="cuda") + 0.5
            outs = []
            for v in (2, 0):
                lib.uwir_conv_variant(v)
                out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                ops.conv_gemm(A, W, out, B=1, H=1, W=M, C_=K, N=N, bias=bias, col_scale=cs)
                torch.cuda.synchronize()
                outs.append(out.float())
            ref = ((A[:4096].float() @ W.float().t()) + bias) * cs
            print(f"M{M} N{N} K{K}: wide vs 128x128 max diff {float((outs[0] - outs[1]).abs().max()):.3e}, "
                  f"wide vs fp32 ref (4096 rows) {float((outs[1][:4096] - ref).abs().max()):.3e}, "
                  f"tail rows equal {bool(torch.equal(outs[0][-300:], outs[1][-300:]))} (ref scale {float(ref.abs().max()):.2f})")
            for v in (2, 0):
                lib.uwir_conv_variant(v)
                run(M, N, K)
        lib.uwir_conv_variant(0)
    elif len(sys.argv) > 1 and sys.argv[1] == "student":
        f32, bf = torch.float32, torch.bfloat16
        for M, N, K, a, o, w, r in ((262144, 64, 64, f32, f32, f32, False), (65536, 128, 128, f32, f32, f32, False),
                                    (16384, 256, 256, f32, f32, f32, False), (65536, 1152, 384, bf, bf, bf, False),
                                    (262144, 64, 192, bf, f32, bf, False), (262144, 192, 384, bf, f32, bf, True),
                                    (262144, 576, 192, bf, bf, bf, False), (262144, 192, 192, bf, f32, bf, True),
                                    (262144, 384, 192, bf, bf, bf, False), (65536, 384, 768, bf, f32, bf, True),
                                    (262144, 64, 64, bf, f32, bf, False), (262144, 128, 192, f32, bf, bf, False),
                                    (262144, 64, 144, f32, bf, bf, False), (262144, 128, 256, f32, f32, bf, False),
                                    (262144, 128, 256, f32, f32, f32, False),
                                    (1048576, 64, 128, f32, f32, bf, False), (1048576, 64, 128, f32, f32, f32, False),
                                    (262144, 256, 128, f32, f32, bf, False), (16384, 2304, 768, bf, bf, bf, False)):
            run(M, N, K, odt=o, adt=a, res=r, wdt=w)
    elif len(sys.argv) > 3:
        M, N, K = (int(v) for v in sys.argv[1:4])
        run(M, N, K, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        for K in (64, 192, 384, 768, 1728):
            run(262144, 384, K)
        run(262144, 384, 192, act=2)
        run(262144, 576, 192)
        run(262144, 192, 192, odt=torch.float32, res=True)
        run(262144, 192, 384, odt=torch.float32, res=True)
        run(262144, 128, 192)
        run(262144, 64, 192, odt=torch.float32)
        run(262144, 576, 1728)
        run(65536, 1152, 3456)
        run(16384, 2304, 6912)
        run(8192, 4096, 4096)
        run(262144, 576, 192, adt=torch.float32)
