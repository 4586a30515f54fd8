"""Randomised shapes for the two launchers with the most dispatch paths (uwir_flash_attn, uwir_conv_gemm 1x1) against fp64
references on the device.  python tools/fuzz_ops.py [cases] [seed]"""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def attn_ref(q, k, vs, b, n, heads, hd):
    outs = [torch.empty(b * n, heads * hd, dtype=torch.float64, device="cuda") for _ in vs]
    for bi in range(b):
        rows = slice(bi * n, (bi + 1) * n)
        for h in range(heads):
            cols = slice(h * hd, (h + 1) * hd)
            a = torch.softmax(q[rows, cols].double() @ k[rows, cols].double().T / math.sqrt(hd), dim=-1)
            for o, v in zip(outs, vs):
                o[rows, cols] = a @ v[rows, cols].double()
    return outs


bad = 0
for ci in range(cases):
    hd = rng.choice([16, 16, 32, 32, 64, 8, 24])
    nv = rng.choice([1, 2])
    heads = rng.choice([1, 3, 4, 8, 12, 16, 32])
    n = rng.choice([rng.randint(1, 300), rng.randint(256, 2100), rng.randint(1024, 5000), 1024, 2048, 4096])
    b = rng.choice([1, 2, 3])
    if b * heads * n * n * hd > 6e11:
        b, heads = 1, min(heads, 8)
    amp = rng.choice([0.3, 1.0, 2.5, 6.0])
    shift = rng.choice([0.0, 0.0, 2.0])      # common offset of the keys: scores of one sign
    dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    w = heads * hd
    q = (torch.randn(b * n, w, generator=g) * amp).to(dt).cuda()
    k = (torch.randn(b * n, w, generator=g) * amp - shift * torch.sign(q.cpu().float()).mean(0, keepdim=True)).to(dt).cuda()
    v1 = torch.randn(b * n, w, generator=g).to(dt).cuda()
    v2 = torch.randn(b * n, w, generator=g).to(dt).cuda() if nv == 2 else None
    vs = [v1] + ([v2] if nv == 2 else [])
    refs = attn_ref(q, k, vs, b, n, heads, hd)
    o = [torch.full((b * n, w), float("nan"), device="cuda") for _ in vs]
    ops.flash_attn(q, k, v1, v2, b, n, heads, hd, 1 / math.sqrt(hd), o[0], o[1] if nv == 2 else None)
    torch.cuda.synchronize()
    tol = 3e-2 if dt == torch.bfloat16 else 2e-5
    for got, ref in zip(o, refs):
        e = rel(got, ref) if bool(torch.isfinite(got).all()) else float("inf")
        if not e < tol:
            bad += 1
            print(f"ATTN MISMATCH case {ci}: b{b} n{n} heads{heads} hd{hd} nv{nv} amp{amp} shift{shift} {dt}: rel {e:.3e}", flush=True)
print(f"flash_attn: {cases} cases, {bad} bad", flush=True)

bad2 = 0
for ci in range(cases):
    M = rng.choice([rng.randint(1, 700), rng.randint(100, 9000), 4096, 8192, 128 * rng.randint(1, 300)])
    K = rng.choice([8, 24, 64, 72, 128, 192, 200, 384, 576, 1024, 1728, 2048])
    N = rng.choice([4, 12, 64, 96, 128, 192, 200, 256, 384, 576, 1152])
    a_dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    o_dt = rng.choice([torch.bfloat16, torch.float32])
    A = torch.randn(M, K, generator=g).bfloat16().to(a_dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda() if rng.random() < 0.7 else None
    cs = (torch.rand(N, generator=g) + 0.5).cuda() if rng.random() < 0.3 else None
    res = torch.randn(M, N, generator=g).to(rng.choice([torch.bfloat16, torch.float32])).cuda() if rng.random() < 0.4 else None
    act = rng.choice([ops.ACT_NONE, ops.ACT_NONE, ops.ACT_GELU, ops.ACT_RELU, ops.ACT_SILU])
    ref = A.double() @ W.double().T
    if bias is not None:
        ref = ref + bias.double()
    if act == ops.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    elif act == ops.ACT_RELU:
        ref = torch.relu(ref)
    elif act == ops.ACT_SILU:
        ref = torch.nn.functional.silu(ref)
    if cs is not None:
        ref = ref * cs.double()
    if res is not None:
        ref = ref + res.double()
    out = torch.full((M, N + 8), float("nan"), device="cuda", dtype=o_dt)
    try:
        ops.conv_gemm(A, W, out[:, :N], B=1, H=1, W=M, C_=K, N=N, bias=bias, col_scale=cs, res=res, act=act)
    except Exception as ex:   # argument checks (alignment of odd N / K) are fine; report them once
        print(f"GEMM case {ci} rejected: M{M} K{K} N{N}: {str(ex)[:80]}")
        continue
    torch.cuda.synchronize()
    got = out[:, :N]
    e = rel(got, ref) if bool(torch.isfinite(got.float()).all()) else float("inf")
    tol = 6e-3 if o_dt == torch.bfloat16 else 5e-4
    pad_ok = bool(torch.isnan(out[:, N:].float()).all())
    if not e < tol or not pad_ok:
        bad2 += 1
        print(f"GEMM MISMATCH case {ci}: M{M} K{K} N{N} A:{a_dt} out:{o_dt} bias{bias is not None} cs{cs is not None} "
              f"res{None if res is None else res.dtype} act{act}: rel {e:.3e} pad_ok {pad_ok}", flush=True)
print(f"conv_gemm 1x1: {cases} cases, {bad2} bad", flush=True)
sys.exit(1 if bad or bad2 else 0)
