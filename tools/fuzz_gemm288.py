"""Randomised shapes for the 256 x 288 GEMM kernel (csrc/gemm_wide288.hip) with shared and per-image weight panels / bias
rows, short and long K, ragged M, strided A / out, against fp64 on the device.   python tools/fuzz_gemm288.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bf = torch.bfloat16
bad = 0
for case in range(cases):
    batched = rng.random() < 0.5
    N = 288 * rng.choice([1, 2, 3, 4])
    K = 64 * rng.choice([3, 4, 6, 9, 12, 15, 16, 27, 40])
    if batched:
        B, rpi = rng.choice([2, 3, 5, 16]), 256 * rng.choice([1, 2, 5, 16])
        while B * rpi < 4096:
            B += 1
        M = B * rpi
    else:
        B, M = 1, rng.choice([4096, 4173, 5000, 8192 + 77, 16384])
        rpi = M
    g = torch.Generator(device="cuda").manual_seed(case)
    pa, po = rng.choice([0, 8]), rng.choice([0, 8])
    xa = torch.randn(M, K + pa, device="cuda", generator=g).to(bf)
    x = xa[:, :K]
    outa = torch.full((M, N + po), 3.0, device="cuda", dtype=bf)
    out = outa[:, :N]
    act = rng.choice([ops.ACT_NONE, ops.ACT_GELU])
    cs = torch.rand(N, device="cuda", generator=g) + 0.5 if rng.random() < 0.5 else None
    if batched:
        w = (torch.randn(B, N, K, device="cuda", generator=g) / K ** 0.5).to(bf)
        bb = torch.randn(B, N, device="cuda", generator=g)
        bias = torch.randn(N, device="cuda", generator=g) if rng.random() < 0.3 else None
        ops.conv_gemm(x, w, out, B=B, H=rpi // 16, W=16, C_=K, N=N, bias=bias, batch_bias=bb, act=act, col_scale=cs, w_batched=True)
        ref = torch.einsum("bnk,bok->bno", x.double().reshape(B, rpi, K), w.double()) + bb.double()[:, None]
        if bias is not None:
            ref = ref + bias.double()
        ref = ref.reshape(M, N)
    else:
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(bf)
        bias = torch.randn(N, device="cuda", generator=g)
        ops.conv_gemm(x, w, out, B=1, H=1, W=M, C_=K, N=N, bias=bias, act=act, col_scale=cs)
        ref = x.double() @ w.double().t() + bias.double()
    if act == ops.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    if cs is not None:
        ref = ref * cs.double()
    err = float((out.double() - ref).abs().max() / (ref.abs().max() + 1e-9))
    pad_ok = po == 0 or float(outa[:, N:].float().min()) == 3.0 == float(outa[:, N:].float().max())
    if err > 6e-3 or not pad_ok:
        bad += 1
        print(f"BAD case {case}: batched={batched} M={M} N={N} K={K} rpi={rpi} act={act} err={err:.3e} pad_ok={pad_ok}")
print(f"gemm 256x288: {cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
