"""Micro-benchmark of uwir_conv_gemm on 1x1 shapes: python tools/bench_gemm.py M N K [act] [out=bf16|f32] [a=bf16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402


def run(M, N, K, act=0, odt=torch.bfloat16, adt=torch.bfloat16, res=False, reps=20):
    A = torch.randn(M, K, device="cuda").to(adt)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
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
    print(f"M{M} N{N} K{K} act{act} out={str(odt)[6:]} a={str(adt)[6:]} res={res}: {ms:.3f} ms  "
          f"{2.0 * M * N * K / ms / 1e9:7.0f} TF/s  {by / ms / 1e6:6.0f} GB/s")


if __name__ == "__main__":
    if len(sys.argv) > 3:
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
