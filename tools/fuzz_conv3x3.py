"""Randomised 3x3 / stride-1 bf16 convolutions (the register-staged halo kernel, the LDS-DMA kernels with their ring /
tile shapes, the resident-weight kernel) against F.conv2d in fp64, each case run three times: the results must also be
bit-identical between runs (a counted-wait race shows up as a rare differing tile).
    python tools/fuzz_conv3x3.py [cases] [seed]"""
import math
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import _lib, ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(3)
lib = _lib.load()
bad = 0
for ci in range(cases):
    C = rng.choice([64, 64, 128, 192, 256, 384, 512, 32, 96])
    N = rng.choice([64, 64, 128, 256, 512, 192])
    B = rng.choice([1, 2, 3, 8])
    H, W = rng.choice([(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (rng.randint(3, 70), rng.randint(3, 70)), (40, 24), (17, 130)])
    if B * H * W * C * N * 9 > 4e10:
        B = 1
    ups = rng.choice([0, 0, 0, 1]) if H * W <= 64 * 64 else 0
    variant = rng.choice([0, 0, 0, 8, 16, 24, 32, 40, 56])     # bits 3-5: DMA conv off / forced shapes / auto without resw
    a_dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    o_dt = rng.choice([torch.bfloat16, torch.float32])
    x = torch.randn(B, C, H, W, generator=g)
    wt = (torch.randn(N, C, 3, 3, generator=g) / math.sqrt(9 * C)).bfloat16()
    bias = torch.randn(N, generator=g)
    act = rng.choice([ops.ACT_NONE, ops.ACT_SILU])
    OH, OW = H << ups, W << ups
    use_res = rng.random() < 0.5 and act == ops.ACT_NONE
    res = torch.randn(B * OH * OW, N, generator=g).to(rng.choice([torch.bfloat16, torch.float32])).cuda() if use_res else None
    xin = x.permute(0, 2, 3, 1).reshape(B * H * W, C).to(a_dt).cuda().contiguous()
    wd = wt.permute(0, 2, 3, 1).reshape(N, -1).cuda().contiguous()
    xe = xin.double().reshape(B, H, W, C).permute(0, 3, 1, 2)
    if ups:
        xe = F.interpolate(xe, scale_factor=2, mode="nearest")
    ref = F.conv2d(xe, wt.double().cuda(), bias.double().cuda(), padding=1)
    if act == ops.ACT_SILU:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(B * OH * OW, N)
    if res is not None:
        ref = ref + res.double()
    outs = []
    lib.uwir_conv_variant(variant)
    try:
        for rep in range(3):
            out = torch.full((B * OH * OW, N), float("nan"), device="cuda", dtype=o_dt)
            ops.conv_gemm(xin, wd, out, B=B, H=H, W=W, C_=C, N=N, KH=3, KW=3, pad=1, ups=ups, bias=bias.cuda(), act=act, res=res)
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.uwir_conv_variant(0)
    e = float((outs[0].double() - ref).norm() / ref.norm()) if bool(torch.isfinite(outs[0].float()).all()) else float("inf")
    same = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    if not e < 1e-2 or not same:
        bad += 1
        print(f"CONV MISMATCH case {ci}: B{B} {H}x{W} C{C}->N{N} ups{ups} variant{variant} A:{a_dt} out:{o_dt} act{act} res{None if res is None else res.dtype}: "
              f"rel {e:.3e} deterministic {same}", flush=True)
print(f"conv3x3: {cases} cases, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
