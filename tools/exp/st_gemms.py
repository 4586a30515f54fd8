"""SpatialTransformer GEMMs of the IR-SDE UNet (M = 8192 rows): A/B/A/B over uwir_conv_variant values."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib, ops  # noqa: E402

f32, bf = torch.float32, torch.bfloat16
SHAPES = ((8192, 512, 2048, bf, f32, True), (8192, 512, 512, bf, f32, True), (8192, 1536, 512, bf, bf, False),
          (8192, 4096, 512, bf, bf, False), (8192, 256, 1024, bf, f32, True), (8192, 256, 256, bf, f32, True),
          (8192, 768, 256, bf, bf, False), (8192, 2048, 256, bf, bf, False), (8192, 512, 512, f32, f32, True),
          (8192, 512, 768, bf, bf, False), (32768, 256, 384, bf, bf, False), (8192, 256, 2048, bf, bf, False))


def timeit(f, reps=50):
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


lib = _lib.load()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 64]
a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
for _ in range(200):
    a @ b
for M, N, K, adt, odt, res in SHAPES:
    A = torch.randn(M, K, device="cuda").to(adt)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=odt)
    r = torch.randn(M, N, device="cuda") if res else None
    f = lambda: ops.conv_gemm(A, W, out, B=8, H=1, W=M // 8, C_=K, N=N, bias=bias, res=r)  # noqa: E731
    t = {v: [] for v in variants}
    outs = {}
    for rnd in range(2):
        for v in variants:
            lib.uwir_conv_variant(v)
            if rnd == 0:
                f()
                outs[v] = out.float().clone()
            t[v].append(timeit(f))
    lib.uwir_conv_variant(0)
    print(f"M{M} N{N} K{K} a={str(adt)[6:]} out={str(odt)[6:]} res={int(res)}: " +
          "  ".join(f"v{v}: " + "/".join(f"{x:.1f}" for x in t[v]) + f" us ({2.0 * M * N * K / min(t[v]) / 1e6:.0f} TF/s)" for v in variants) +
          f"  max diff {max(float((outs[v] - outs[variants[0]]).abs().max()) for v in variants):.1e}", flush=True)
