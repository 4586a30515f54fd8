"""GPU parity tests, one per C-ABI entry point: HIP kernel vs the CPU oracle / plain torch fp64 math."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import student_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd import _lib, ops as o
    _lib.load()  # fail loudly if the HIP library is missing
    return o


def to_rows(t):  # NCHW cpu -> [B*H*W, C] cuda fp32
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c).contiguous().float().cuda()


def from_rows(t, b, h, w):  # [B*H*W, C] cuda -> NCHW cpu fp64
    return t.float().reshape(b, h, w, -1).permute(0, 3, 1, 2).double().cpu()


def err(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def rel(a, b):
    return err(a, b) / max(float(b.double().abs().max()), 1e-12)


def test_dwt_idwt(ops):
    g = torch.Generator().manual_seed(1)
    for (b, c, h, w) in ((2, 8, 8, 12), (1, 4, 5, 6), (1, 64, 32, 32)):
        x = torch.randn(b, c, h, w, generator=g)
        ll, lh, hl, hh = R.haar_dwt2(x.double())
        h2, w2 = h // 2, w // 2
        dll = torch.empty(b * h2 * w2, c, device="cuda")
        dhf = torch.empty(b * h2 * w2, 3 * c, device="cuda")
        ops.dwt2(to_rows(x), b, h, w, c, dll, dhf)
        assert err(from_rows(dll, b, h2, w2), ll) < 1e-6
        assert err(from_rows(dhf, b, h2, w2), torch.cat([lh, hl, hh], 1)) < 1e-6
        out = torch.empty(b * 4 * h2 * w2, c, device="cuda")
        ops.idwt2(dll, dhf, b, h2, w2, c, out)
        assert err(from_rows(out, b, 2 * h2, 2 * w2), R.haar_idwt2(ll, lh, hl, hh)) < 1e-6
    # strided destination (concat buffer): only the addressed columns may change
    x = torch.randn(1, 4, 4, 4, generator=g)
    big = torch.full((16, 8), 7.0, device="cuda")
    ll, lh, hl, hh = R.haar_dwt2(x.double())
    dll, dhf = to_rows(ll.float()), to_rows(torch.cat([lh, hl, hh], 1).float())
    ops.idwt2(dll, dhf, 1, 2, 2, 4, big[:, 4:])
    assert float(big[:, :4].min()) == 7.0 and err(from_rows(big[:, 4:], 1, 4, 4), x.double()) < 1e-6


@pytest.mark.parametrize("shape", [(2, 48, 4, 4), (1, 192, 16, 16), (2, 64, 40, 24), (1, 768, 1, 1)])
def test_ln2d_affine(ops, shape):
    b, c, h, w = shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(shape, generator=g) * 3 + 5
    sd = {"norm.weight": torch.randn(1, c, 1, 1, generator=g), "norm.bias": torch.randn(1, c, 1, 1, generator=g),
          "linear.weight": torch.randn(2 * c, 7, generator=g) * 0.2, "linear.bias": torch.randn(2 * c, generator=g)}
    cond = torch.randn(b, 7, generator=g)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = R.ada_layernorm2d(sd64, "", x.double(), cond.double())
    ss = F.linear(cond, sd["linear.weight"], sd["linear.bias"]).cuda()
    xr = to_rows(x)
    n = h * w
    a, s, mean, pooled = (torch.empty(b, c, device="cuda") for _ in range(4))
    ws = torch.empty(max(ops.ln2d_workspace_bytes(b, n, c) // 8, 1), dtype=torch.float64, device="cuda")
    ops.ln2d_affine(xr, b, n, c, 1e-5, sd["norm.weight"].reshape(-1).cuda(), sd["norm.bias"].reshape(-1).cuda(), ss,
                    a, s, mean, pooled, ws)
    got = xr.reshape(b, n, c) * a[:, None, :] + s[:, None, :]
    tol = 2e-4 if n == 1 else 2e-5
    assert rel(from_rows(got.reshape(-1, c), b, h, w), ref) < tol
    assert err(mean.cpu(), x.double().mean(dim=(2, 3))) < 1e-5
    assert err(pooled.cpu(), ref.mean(dim=(2, 3))) < 1e-4
    # plain LayerNorm2d (no cond), statistics-only mode
    a2, s2 = torch.empty(b, c, device="cuda"), torch.empty(b, c, device="cuda")
    ops.ln2d_affine(xr, b, n, c, 1e-5, sd["norm.weight"].reshape(-1).cuda(), sd["norm.bias"].reshape(-1).cuda(), None,
                    a2, s2, None, None, ws)
    got = xr.reshape(b, n, c) * a2[:, None, :] + s2[:, None, :]
    assert rel(from_rows(got.reshape(-1, c), b, h, w), R.layernorm2d(sd64, "norm.", x.double())) < tol
    m2 = torch.empty(b, c, device="cuda")
    ops.ln2d_affine(xr, b, n, c, 1e-5, None, None, None, None, None, m2, None, ws)
    assert err(m2.cpu(), x.double().mean(dim=(2, 3))) < 1e-5


def test_linear_and_fold(ops):
    g = torch.Generator().manual_seed(3)
    x, w, bias = torch.randn(3, 513, generator=g), torch.randn(37, 513, generator=g) * 0.05, torch.randn(37, generator=g)
    for act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_RELU, F.relu), (ops.ACT_GELU, F.gelu),
                    (ops.ACT_SIGMOID, torch.sigmoid), (ops.ACT_SILU, F.silu)):
        y = torch.empty(3, 37, device="cuda")
        ops.linear(x.cuda(), w.cuda(), bias.cuda(), y, act)
        assert err(y, fn(F.linear(x.double(), w.double(), bias.double()))) < 2e-5
    # more rows than one pass holds, long feature vectors (fewer rows per pass), a column slice of a wider matrix as x
    for (b, i, o) in ((40, 512, 70), (5, 3000, 9), (16, 768, 1500), (19, 512, 2051), (2, 12288, 3)):   # 3rd, 4th: wide kernel
        xw = torch.randn(b, i + 8, generator=g).cuda()
        w2, b2 = (torch.randn(o, i, generator=g) * i ** -0.5).cuda(), torch.randn(o, generator=g).cuda()
        y2 = torch.full((b, o + 3), 7.0, device="cuda")
        ops.linear(xw[:, 4:4 + i], w2, b2, y2[:, :o])
        assert err(y2[:, :o], F.linear(xw[:, 4:4 + i].double(), w2.double(), b2.double())) < 2e-5
        assert bool((y2[:, o:] == 7.0).all())
    a, s, cw = (torch.randn(2, 12, generator=g).cuda() for _ in range(3))
    os_, oh = torch.empty_like(a), torch.empty_like(a)
    ops.chan_attn_fold(a, s, cw, os_, oh)
    assert err(os_, 1 + a * cw) < 1e-6 and err(oh, s * cw) < 1e-6


def _wdt(prec):
    return torch.bfloat16 if prec == "bf16" else torch.float32


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 5, 7, 48, 144), (1, 16, 16, 192, 576), (3, 9, 11, 64, 64), (1, 2, 2, 1728, 200)])
def test_conv_gemm_1x1_epilogues(ops, prec, shape):
    b, h, w, k, n = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b, k, h, w, generator=g)
    wt = torch.randn(n, k, generator=g) / math.sqrt(k)
    bias, bb = torch.randn(n, generator=g), torch.randn(b, n, generator=g)
    asc, ash = torch.rand(b, k, generator=g) + 0.5, torch.randn(b, k, generator=g)
    res = torch.randn(b, n, h, w, generator=g)
    rsc, rsh = torch.rand(b, n, generator=g) + 0.5, torch.randn(b, n, generator=g)
    tol = 2e-5 if prec == "fp32" else 3e-2
    wd = wt.to(_wdt(prec)).cuda()
    xr = to_rows(x)
    # (1) affine prologue + bias + per-sample bias + GELU
    xa = x.double() * asc[:, :, None, None].double() + ash[:, :, None, None].double()
    ref = F.gelu(F.conv2d(xa, wt.double()[:, :, None, None], bias.double()) + bb.double()[:, :, None, None])
    out = torch.empty(b * h * w, n, device="cuda")
    ops.conv_gemm(xr, wd, out, B=b, H=h, W=w, C_=k, N=n, a_scale=asc.cuda(), a_shift=ash.cuda(), bias=bias.cuda(),
                  batch_bias=bb.cuda(), act=ops.ACT_GELU)
    assert rel(from_rows(out, b, h, w), ref) < tol
    # (2) column scale + affine residual, bf16 / strided output
    cs = torch.rand(n, generator=g) + 0.5
    ref2 = F.conv2d(x.double(), wt.double()[:, :, None, None], bias.double()) * cs.double()[None, :, None, None] \
        + res.double() * rsc.double()[:, :, None, None] + rsh.double()[:, :, None, None]
    big = torch.zeros(b * h * w, n + 8, device="cuda", dtype=_wdt(prec))
    ops.conv_gemm(xr, wd, big[:, 8:], B=b, H=h, W=w, C_=k, N=n, bias=bias.cuda(), res=to_rows(res), res_scale=rsc.cuda(),
                  res_shift=rsh.cuda(), col_scale=cs.cuda())
    assert rel(from_rows(big[:, 8:], b, h, w), ref2) < (tol if prec == "fp32" else 4e-2)
    assert float(big[:, :8].float().abs().max()) == 0.0
    # (3) sigmoid + gate blend
    t, f = torch.randn(b, n, h, w, generator=g), torch.randn(b, n, h, w, generator=g)
    al = torch.sigmoid(F.conv2d(x.double(), wt.double()[:, :, None, None], bias.double()))
    tf = torch.cat([to_rows(t), to_rows(f)], dim=1)
    alpha, merged = torch.empty(b * h * w, n, device="cuda"), torch.empty(b * h * w, n, device="cuda")
    ops.conv_gemm(xr, wd, alpha, B=b, H=h, W=w, C_=k, N=n, bias=bias.cuda(), act=ops.ACT_SIGMOID, blend_t=tf[:, :n],
                  blend_f=tf[:, n:], blend_out=merged)
    assert err(from_rows(alpha, b, h, w), al) < (1e-5 if prec == "fp32" else 2e-2)
    assert rel(from_rows(merged, b, h, w), al * t.double() + (1 - al) * f.double()) < (1e-5 if prec == "fp32" else 2e-2)
    if prec == "bf16":  # bf16 activations as the A operand
        out2 = torch.empty(b * h * w, n, device="cuda")
        ops.conv_gemm(xr.to(torch.bfloat16), wd, out2, B=b, H=h, W=w, C_=k, N=n, bias=bias.cuda())
        assert rel(from_rows(out2, b, h, w), F.conv2d(x.double(), wt.double()[:, :, None, None], bias.double())) < tol


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_conv_gemm_spatial(ops, prec):
    g = torch.Generator().manual_seed(5)
    tol = 2e-5 if prec == "fp32" else 3e-2
    b, c, h, w, n = 2, 64, 12, 8, 128
    x = torch.randn(b, c, h, w, generator=g)
    xr = to_rows(x)
    # Conv2d(k=2, s=2)  (S/model.py:93-96)
    wt = torch.randn(n, c, 2, 2, generator=g) / math.sqrt(4 * c)
    bias = torch.randn(n, generator=g)
    out = torch.empty(b * (h // 2) * (w // 2), n, device="cuda")
    ops.conv_gemm(xr, wt.permute(0, 2, 3, 1).reshape(n, -1).to(_wdt(prec)).cuda().contiguous(), out, B=b, H=h, W=w, C_=c,
                  N=n, KH=2, KW=2, stride=2, pad=0, bias=bias.cuda())
    assert rel(from_rows(out, b, h // 2, w // 2), F.conv2d(x.double(), wt.double(), bias.double(), stride=2)) < tol
    # Conv2d(k=3, p=1) with normalise-on-load: padding must stay zero AFTER the affine
    wt3 = torch.randn(n, c, 3, 3, generator=g) / math.sqrt(9 * c)
    asc, ash = torch.rand(b, c, generator=g) + 0.5, torch.randn(b, c, generator=g)
    xa = x.double() * asc.double()[:, :, None, None] + ash.double()[:, :, None, None]
    out3 = torch.empty(b * h * w, n, device="cuda")
    ops.conv_gemm(xr, wt3.permute(0, 2, 3, 1).reshape(n, -1).to(_wdt(prec)).cuda().contiguous(), out3, B=b, H=h, W=w,
                  C_=c, N=n, KH=3, KW=3, stride=1, pad=1, a_scale=asc.cuda(), a_shift=ash.cuda(), act=ops.ACT_SILU)
    assert rel(from_rows(out3, b, h, w), F.silu(F.conv2d(xa, wt3.double(), padding=1))) < tol
    # Conv2d(k=4, s=2, p=1)  (IR-SDE UNet downsample)
    wt4 = torch.randn(n, c, 4, 4, generator=g) / math.sqrt(16 * c)
    out4 = torch.empty(b * (h // 2) * (w // 2), n, device="cuda")
    ops.conv_gemm(xr, wt4.permute(0, 2, 3, 1).reshape(n, -1).to(_wdt(prec)).cuda().contiguous(), out4, B=b, H=h, W=w,
                  C_=c, N=n, KH=4, KW=4, stride=2, pad=1)
    assert rel(from_rows(out4, b, h // 2, w // 2), F.conv2d(x.double(), wt4.double(), stride=2, padding=1)) < tol
    # ConvTranspose2d(k=2, s=2) as GEMM + depth-to-space into the left half of a concat buffer (S/model.py:98-104)
    co = 32
    wtt = torch.randn(c, co, 2, 2, generator=g) / math.sqrt(c)
    bt = torch.randn(co, generator=g)
    cat = torch.full((b * 4 * h * w, 2 * co), 3.0, device="cuda")
    ops.conv_gemm(xr, wtt.permute(2, 3, 1, 0).reshape(4 * co, c).to(_wdt(prec)).cuda().contiguous(), cat[:, :co], B=b, H=h,
                  W=w, C_=c, N=4 * co, bias=bt.repeat(4).cuda(), store_mode=1)
    assert rel(from_rows(cat[:, :co], b, 2 * h, 2 * w), F.conv_transpose2d(x.double(), wtt.double(), bt.double(), stride=2)) < tol
    assert float(cat[:, co:].min()) == 3.0


@pytest.mark.parametrize("shape", [(800, 768, 3072), (800, 768, 768), (130, 200, 1024), (16, 512, 512)])
def test_conv_gemm_splitk(ops, shape):
    """Plain bf16 GEMMs with fewer output tiles than CUs cut K into slices (workspace lent by ops.conv_gemm): same
    result as the single-slice launch up to fp32 summation order, bit-identical run to run (slice-ordered sum, no
    atomics on data), counters re-armed for the next call."""
    M, N, K = shape
    g = torch.Generator().manual_seed(31)
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    bias, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    ref = A.double() @ Wt.double().T + bias.double() + res.double()
    outs = []
    for split in (True, True, False):
        ops._SPLITK = split
        try:
            o = torch.empty(M, N, device="cuda")
            ops.gemm(A, Wt, o, bias=bias, res=res)
        finally:
            ops._SPLITK = True
        assert err(o, ref) < 2e-4
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - outs[2]).abs().max()) < 1e-4
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)      # GELU epilogue on the combined sum, bf16 store
    ops.gemm(A, Wt, ob, bias=bias, act=ops.ACT_GELU)
    assert err(ob, torch.nn.functional.gelu(A.double() @ Wt.double().T + bias.double())) < 3e-2


@pytest.mark.parametrize("geom", [(3, 16, 24, 192, 128), (1, 10, 13, 200, 192), (2, 8, 16, 576, 64), (1, 16, 17, 72, 64)])
def test_conv_gemm_epilogue_classes(ops, geom):
    """The compile-time epilogue classes of the 1x1 GEMM kernels (plain: bias / per-sample bias / activation / column
    scale with the bias as accumulator start; residual: + fp32 residual, plain or affine, also as start value; full:
    everything else) against one fp64 formula, on shapes with ragged M and N tiles, images that do and do not fill
    whole 128-row tiles (the affine residual falls back to the full class when they do not), bf16 and fp32 inputs."""
    b, h, w, n, k = geom
    m = b * h * w
    g = torch.Generator().manual_seed(77)
    wt = (torch.randn(n, k, generator=g) / k ** 0.5).bfloat16().cuda()
    bias, cs = torch.randn(n, generator=g).cuda(), (torch.rand(n, generator=g) + 0.5).cuda()
    bb = torch.randn(b, n, generator=g).cuda()
    res = torch.randn(m, n, generator=g).cuda()
    rs, rh = (torch.rand(b, n, generator=g) + 0.5).cuda(), torch.randn(b, n, generator=g).cuda()
    img = torch.arange(m, device="cuda") // (h * w)
    for a_dt in (torch.bfloat16, torch.float32):
        a = torch.randn(m, k, generator=g).bfloat16().to(a_dt).cuda()      # bf16-representable either way
        acc = a.double() @ wt.double().T
        cases = {
            "lean": (dict(bias=bias, batch_bias=bb, act=ops.ACT_GELU, col_scale=cs),
                     torch.nn.functional.gelu(acc + bias.double() + bb.double()[img]) * cs.double()),
            "lean_nobias": (dict(), acc),
            "res": (dict(bias=bias, res=res), acc + bias.double() + res.double()),
            "res_affine": (dict(bias=bias, batch_bias=bb, res=res, res_scale=rs, res_shift=rh),
                           acc + bias.double() + bb.double()[img] + res.double() * rs.double()[img] + rh.double()[img]),
            "full_act_res": (dict(bias=bias, act=ops.ACT_RELU, res=res), torch.relu(acc + bias.double()) + res.double()),
        }
        for name, (kw, want) in cases.items():
            for odt, tol in ((torch.float32, 2e-4), (torch.bfloat16, 2.0 ** -8 * float(want.abs().max()))):   # half a bf16 ulp
                out = torch.full((m, n), float("nan"), device="cuda", dtype=odt)
                ops.conv_gemm(a, wt, out, B=b, H=h, W=w, C_=k, N=n, **kw)
                assert err(out, want) < tol, (name, a_dt, odt)


def test_conv_gemm_rejects_bad_arguments(ops):
    x = torch.zeros(16, 12, device="cuda")
    w = torch.zeros(8, 12, device="cuda")
    with pytest.raises(ValueError):
        ops.conv_gemm(x, w.to(torch.bfloat16), torch.empty(16, 8, device="cuda"), B=1, H=4, W=4, C_=12, N=8)  # C % 8
    with pytest.raises(ValueError):
        ops.conv_gemm(x, w, torch.empty(16, 4, device="cuda"), B=1, H=4, W=4, C_=12, N=8)  # ldo < N


@pytest.mark.parametrize("io", [("fp32", "fp32"), ("fp32", "bf16"), ("bf16", "bf16")])
@pytest.mark.parametrize("shape", [(2, 48, 9, 21), (1, 576, 16, 16), (1, 20, 3, 2), (2, 64, 40, 8), (1, 40, 17, 33)])
def test_dwconv_ms(ops, io, shape):
    b, c, h, w = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn(shape, generator=g)
    ws = [torch.randn(c, 1, k, k, generator=g) / k for k in (3, 5, 7)]
    bs = [torch.randn(c, generator=g) for _ in range(3)]
    asc, ash = torch.rand(b, c, generator=g) + 0.5, torch.randn(b, c, generator=g)
    idt, odt = (torch.bfloat16 if t == "bf16" else torch.float32 for t in io)
    xin = to_rows(x).to(idt)
    xe = xin.float().cpu().reshape(b, h, w, c).permute(0, 3, 1, 2).double()
    taps = tuple(wk.reshape(c, -1).t().contiguous().cuda() for wk in ws)
    for use_aff in (False, True):
        xa = xe * asc.double()[:, :, None, None] + ash.double()[:, :, None, None] if use_aff else xe
        ref = torch.cat([F.conv2d(xa, wk.double(), bk.double(), padding=wk.shape[-1] // 2, groups=c) for wk, bk in zip(ws, bs)], 1)
        out = torch.empty(b * h * w, 3 * c, device="cuda", dtype=odt)
        ops.dwconv_ms(xin, b, h, w, c, taps, tuple(t.cuda() for t in bs), out,
                      in_scale=asc.cuda() if use_aff else None, in_shift=ash.cuda() if use_aff else None)
        assert rel(from_rows(out, b, h, w), ref) < (1e-5 if odt == torch.float32 else 1e-2)


@pytest.mark.parametrize("io", [("fp32", "bf16"), ("bf16", "bf16")])
@pytest.mark.parametrize("nt", [1, 2, 3, 5])
def test_dwconv_ms_pipelined_tiles(ops, io, nt):
    """The wave-per-quad kernel walks nt tiles per block with the next tile's halo prefetched (bf16 input: loader /
    storer wave roles): every nt, ragged tile counts and clipped edge tiles must give the one-tile result bit for bit."""
    from underwater_image_restoration_amd import _lib
    lib = _lib.load()
    b, c, h, w = 2, 64, 70, 53   # 5 x 4 tiles, clipped on both edges; 20 tiles is not a multiple of 3
    g = torch.Generator().manual_seed(16)
    x = torch.randn(b, c, h, w, generator=g)
    ws = [torch.randn(c, 1, k, k, generator=g) / k for k in (3, 5, 7)]
    bs = [torch.randn(c, generator=g) for _ in range(3)]
    asc, ash = torch.rand(b, c, generator=g) + 0.5, torch.randn(b, c, generator=g)
    idt = torch.bfloat16 if io[0] == "bf16" else torch.float32
    xin = to_rows(x).to(idt)
    xe = xin.float().cpu().reshape(b, h, w, c).permute(0, 3, 1, 2).double()
    taps = tuple(wk.reshape(c, -1).t().contiguous().cuda() for wk in ws)
    for use_aff in (False, True):
        xa = xe * asc.double()[:, :, None, None] + ash.double()[:, :, None, None] if use_aff else xe
        ref = torch.cat([F.conv2d(xa, wk.double(), bk.double(), padding=wk.shape[-1] // 2, groups=c) for wk, bk in zip(ws, bs)], 1)
        outs = []
        for n in (1, nt):
            lib.uwir_dwconv_tiles_per_block(n)
            try:
                out = torch.full((b * h * w, 3 * c), float("nan"), device="cuda", dtype=torch.bfloat16)
                ops.dwconv_ms(xin, b, h, w, c, taps, tuple(t.cuda() for t in bs), out,
                              in_scale=asc.cuda() if use_aff else None, in_shift=ash.cuda() if use_aff else None)
                torch.cuda.synchronize()
            finally:
                lib.uwir_dwconv_tiles_per_block(0)
            outs.append(out)
        assert rel(from_rows(outs[0], b, h, w), ref) < 1e-2
        assert torch.equal(outs[0], outs[1])


def _attn_ref(q, k, vs, heads, hd):
    b, n, _ = q.shape
    qh = q.reshape(b, n, heads, hd).permute(0, 2, 1, 3)
    kh = k.reshape(b, n, heads, hd).permute(0, 2, 1, 3)
    a = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    return [(a @ v.reshape(b, n, heads, hd).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(b, n, heads * hd) for v in vs]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 16, 12, 16, 1), (1, 50, 12, 16, 1), (1, 1000, 3, 32, 1), (2, 200, 2, 64, 1),
                                 (1, 1, 4, 16, 2), (2, 333, 4, 16, 2), (1, 129, 4, 32, 2), (1, 257, 4, 64, 2),
                                 (1, 70, 12, 4, 1), (1, 70, 4, 8, 2), (1, 4096, 2, 16, 1),
                                 (1, 300, 2, 128, 1), (2, 130, 3, 96, 2), (1, 1100, 2, 128, 2)])   # wide heads: hd > 64
def test_flash_attn(ops, prec, cfg):
    b, n, heads, hd, nv = cfg
    g = torch.Generator().manual_seed(7)
    wdt = heads * hd
    dt = _wdt(prec)
    # one wide buffer holding q|k|v (strided views, like the fused qkv GEMM output)
    qkv = (torch.randn(b * n, 3 * wdt + 8, generator=g) * 1.5).to(dt)
    v2 = torch.randn(b * n, wdt, generator=g).to(dt)
    q, k, v = qkv[:, :wdt], qkv[:, wdt:2 * wdt], qkv[:, 2 * wdt:3 * wdt]
    vs = [v] + ([v2] if nv == 2 else [])
    refs = _attn_ref(q.double().reshape(b, n, wdt), k.double().reshape(b, n, wdt), [t.double().reshape(b, n, wdt) for t in vs],
                     heads, hd)
    dq = qkv.cuda()
    dv2 = v2.cuda() if nv == 2 else None
    o = torch.full((b * n, 2 * wdt + 4), -9.0, device="cuda")  # fp32 outputs side by side (gate-input layout)
    ops.flash_attn(dq[:, :wdt], dq[:, wdt:2 * wdt], dq[:, 2 * wdt:3 * wdt], dv2, b, n, heads, hd, 1.0 / math.sqrt(hd),
                   o[:, :wdt], o[:, wdt:2 * wdt] if nv == 2 else None)
    tol = 2e-5 if prec == "fp32" else 3e-2  # bf16: q*scale*log2e is re-rounded to bf16 inside the kernel
    assert err(o[:, :wdt].reshape(b, n, wdt), refs[0]) < tol
    if nv == 2:
        assert err(o[:, wdt:2 * wdt].reshape(b, n, wdt), refs[1]) < tol
    else:
        assert float(o[:, wdt:].max()) == -9.0
    assert float(o[:, 2 * wdt:].max()) == -9.0
    # pre-scaled q (scale == 0 convention): q already carries (1/sqrt(hd))*log2(e), rounded once
    cfac = 1.4426950408889634 / math.sqrt(hd)
    qs = (q.double() * cfac).to(dt)
    refs_ps = _attn_ref(qs.double().reshape(b, n, wdt) / cfac, k.double().reshape(b, n, wdt),
                        [t.double().reshape(b, n, wdt) for t in vs], heads, hd)
    o2 = torch.empty(b * n, wdt, device="cuda")
    ops.flash_attn(qs.cuda(), dq[:, wdt:2 * wdt], dq[:, 2 * wdt:3 * wdt], None, b, n, heads, hd, 0.0, o2, None)
    assert err(o2.reshape(b, n, wdt), refs_ps[0]) < (2e-5 if prec == "fp32" else 2e-2)
    if prec == "bf16":  # bf16 output buffer
        ob = torch.empty(b * n, wdt, device="cuda", dtype=torch.bfloat16)
        ops.flash_attn(dq[:, :wdt], dq[:, wdt:2 * wdt], dq[:, 2 * wdt:3 * wdt], None, b, n, heads, hd, 1.0 / math.sqrt(hd), ob, None)
        assert err(ob.reshape(b, n, wdt), refs[0]) < 4e-2


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_flash_attn_forced_rescale(ops, prec):
    """A key whose score jumps far above the running max late in the sequence forces the deferred-max rescale."""
    b, n, heads, hd = 1, 300, 1, 16
    g = torch.Generator().manual_seed(8)
    q, k, v = (torch.randn(n, hd, generator=g) for _ in range(3))
    k[250] = q[5] * 6.0   # spike for query 5 in the 4th key tile
    k[299] = q[100] * 9.0
    dt = _wdt(prec)
    q, k, v = q.to(dt), k.to(dt), v.to(dt)
    (ref,) = _attn_ref(q.double()[None], k.double()[None], [v.double()[None]], heads, hd)
    o = torch.empty(n, hd, device="cuda")
    ops.flash_attn(q.cuda(), k.cuda(), v.cuda(), None, b, n, heads, hd, 1.0 / math.sqrt(hd), o, None)
    assert err(o[None], ref) < (2e-5 if prec == "fp32" else 2e-2)


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 64, 128, 128), (2, 20, 8, 32), (1, 4, 1, 1), (1, 8, 12, 10), (1, 4, 2, 4),
                                   (1, 8, 68, 120),    # 68 x 120: level-2 map of a 544 x 960 frame: direct DFT (17 | 68) x mixed radix
                                   (2, 8, 30, 40), (1, 20, 135, 240), (1, 4, 90, 6), (1, 8, 270, 480),   # 2^a 3^b 5^c lines
                                   (1, 4, 14, 22),     # 7 | 14, 11 | 22: prime-radix butterflies (round 3)
                                   (1, 8, 112, 56), (2, 4, 28, 26), (1, 4, 49, 154), (1, 4, 13, 7),   # 7^2, 7 * 11 * 2, 13 * 2, L = the prime itself
                                   (1, 4, 19, 34)])    # 19, 17 | 34: primes above 13 keep the direct DFT
def test_fft_roundtrip_and_parts(ops, shape):
    b, c, h, w = shape
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape, generator=g) - 0.3
    spec = torch.fft.fft2(x.double(), norm="ortho")
    im = spec.imag.clone()
    for ky in ({0, h // 2} if h % 2 == 0 else {0}):
        for kx in ({0, w // 2} if w % 2 == 0 else {0}):
            im[..., ky, kx] = 0.0
    xr = to_rows(x)
    amp, ph = torch.empty_like(xr), torch.empty_like(xr)
    ws = torch.empty(ops.fft_workspace_bytes(b, h, w, c) // 4, device="cuda")
    ops.fft2_amp_phase(xr, b, h, w, c, amp, ph, ws)
    ref_amp, ref_ph = torch.complex(spec.real, im).abs(), torch.atan2(im, spec.real)
    assert err(from_rows(amp, b, h, w), ref_amp) < 2e-5 * max(1.0, float(ref_amp.max()))
    # phase is discontinuous at +-pi: compare on the circle, weighted by the amplitude
    dphi = from_rows(ph, b, h, w) - ref_ph
    assert float((torch.abs(torch.sin(dphi / 2)) * ref_amp).max()) < 2e-5 * max(1.0, float(ref_amp.max()))
    got_ph = from_rows(ph, b, h, w)
    for ky in ({0, h // 2} if h % 2 == 0 else {0}):  # self-conjugate bins: exactly 0 or +pi, never -pi
        for kx in ({0, w // 2} if w % 2 == 0 else {0}):
            assert torch.equal(got_ph[..., ky, kx].float(), ref_ph[..., ky, kx].float())
    # inverse with a modulated spectrum + addend
    amp2 = torch.randn(shape, generator=g)
    ph2 = torch.randn(shape, generator=g) * 4
    add = torch.randn(shape, generator=g)
    ref = add.double() + torch.fft.ifft2(torch.complex(amp2.double() * torch.cos(ph2.double()), amp2.double() * torch.sin(ph2.double())),
                                         norm="ortho").real
    out = torch.empty_like(xr)
    ops.ifft2_polar_add(to_rows(amp2), to_rows(ph2), to_rows(add), b, h, w, c, out, ws)
    assert err(from_rows(out, b, h, w), ref) < 2e-5


def test_bilinear(ops):
    g = torch.Generator().manual_seed(10)
    for (hs, ws_, hd, wd) in ((7, 7, 16, 16), (224, 224, 16, 8), (224, 224, 4, 4), (8, 8, 8, 8), (7, 7, 1, 1)):
        src = torch.rand(2, 10, hs, ws_, generator=g)
        ref = F.interpolate(src, size=(hd, wd), mode="bilinear", align_corners=False)
        dst = torch.full((2 * hd * wd, 16), 5.0, device="cuda")
        ops.bilinear_nchw_to_nhwc(src.cuda(), dst[:, 4:14], hd, wd)
        assert err(from_rows(dst[:, 4:14], 2, hd, wd), ref) < 1e-6
        assert float(dst[:, :4].min()) == 5.0 and float(dst[:, 14:].min()) == 5.0


def test_bilinear_rows(ops):
    """The decoder's resize to its skip's size (model.py:150-151) on NHWC rows, fp32 and bf16, strided source and
    destination, against F.interpolate on the CPU (fp32: 1e-6; bf16: inputs exact in bf16, one rounding of the result)."""
    g = torch.Generator().manual_seed(12)
    for (hs, ws_, hd, wd, C) in ((16, 20, 18, 22, 32), (36, 40, 38, 44, 16), (4, 4, 4, 4, 8), (2, 2, 7, 3, 4)):
        src = torch.rand(2, C, hs, ws_, generator=g).bfloat16().float()
        ref = F.interpolate(src, size=(hd, wd), mode="bilinear", align_corners=False)
        for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 4e-3)):
            rows = torch.zeros(2 * hs * ws_, C + 8, device="cuda", dtype=dt)
            rows[:, 4:4 + C] = src.permute(0, 2, 3, 1).reshape(-1, C).to(dt).cuda()
            dst = torch.full((2 * hd * wd, 2 * C), 5.0, device="cuda", dtype=dt)
            ops.bilinear_rows(rows[:, 4:4 + C], 2, hs, ws_, C, dst[:, C:], hd, wd)
            assert err(from_rows(dst[:, C:].float(), 2, hd, wd), ref) < tol
            assert float(dst[:, :C].float().min()) == 5.0
    with pytest.raises(ValueError):
        ops.bilinear_rows(torch.zeros(8, 6, device="cuda"), 2, 2, 2, 6, torch.zeros(8, 6, device="cuda"), 2, 2)


def test_stem_and_head(ops):
    g = torch.Generator().manual_seed(11)
    b, h, w = 2, 12, 20
    x = torch.rand(b, 3, h, w, generator=g)
    ws_, bs_ = torch.randn(64, 3, 3, 3, generator=g) * 0.2, torch.randn(64, generator=g)
    out = torch.empty(b * h * w, 64, device="cuda")
    ops.stem_conv3x3(x.cuda(), ws_.permute(2, 3, 1, 0).contiguous().cuda(), bs_.cuda(), out)
    assert err(from_rows(out, b, h, w), F.conv2d(x.double(), ws_.double(), bs_.double(), padding=1)) < 1e-5
    feat = torch.randn(b, 64, h, w, generator=g)
    wh, bh = torch.randn(3, 64, 3, 3, generator=g) * 0.05, torch.randn(3, generator=g) * 0.1
    asc, ash = torch.rand(b, 64, generator=g) + 0.5, torch.randn(b, 64, generator=g)
    fa = feat.double() * asc.double()[:, :, None, None] + ash.double()[:, :, None, None]
    pre_ref = F.conv2d(fa, wh.double(), bh.double(), padding=1) + x.double()
    o, pre = torch.empty(b, 3, h, w, device="cuda"), torch.empty(b, 3, h, w, device="cuda")
    ops.head_conv3x3(to_rows(feat), b, h, w, 64, asc.cuda(), ash.cuda(), wh.permute(0, 2, 3, 1).contiguous().cuda(), bh.cuda(), 3,
                     x.cuda(), o, pre)
    assert err(pre, pre_ref) < 2e-5 and err(o, pre_ref.clamp(0, 1)) < 2e-5
    # widths: a multiple of 16 takes the LDS-tiled kernel (maps larger than one 8 x 32 tile, ragged edges), others the
    # wave-per-pixel form
    for cin, (hh, ww) in ((48, (19, 70)), (24, (9, 11))):
        feat = torch.randn(b, cin, hh, ww, generator=g)
        xr = torch.rand(b, 2, hh, ww, generator=g)
        wh, bh = torch.randn(2, cin, 3, 3, generator=g) * 0.05, torch.randn(2, generator=g) * 0.1
        ref = F.conv2d(feat.double(), wh.double(), bh.double(), padding=1) + xr.double()
        o, pre = torch.empty(b, 2, hh, ww, device="cuda"), torch.empty(b, 2, hh, ww, device="cuda")
        ops.head_conv3x3(to_rows(feat), b, hh, ww, cin, None, None, wh.permute(0, 2, 3, 1).contiguous().cuda(), bh.cuda(), 2,
                         xr.cuda(), o, pre)
        assert err(pre, ref) < 2e-5 and err(o, ref.clamp(0, 1)) < 2e-5, cin


@pytest.mark.parametrize("cin", [64, 32])
def test_head_mfma_on_bf16_rows(ops, cin):
    """The student's head on bf16 rows (bf16 mode) runs on the matrix cores with hi + lo splits of the normalised input and
    of the weights: against fp64 on the same bf16 rows, and against the per-pixel kernel it replaces (uwir_head_variant)."""
    from underwater_image_restoration_amd import _lib
    g = torch.Generator().manual_seed(111)
    b, h, w = 2, 41, 53                                             # ragged in both directions (16-px strips, 32-row blocks)
    x = torch.rand(b, 3, h, w, generator=g)
    feat = (torch.randn(b, cin, h, w, generator=g) * 3).bfloat16()
    wh, bh = torch.randn(3, cin, 3, 3, generator=g) * 0.05, torch.randn(3, generator=g) * 0.1
    asc, ash = torch.rand(b, cin, generator=g) + 0.5, torch.randn(b, cin, generator=g)
    fa = feat.double() * asc.double()[:, :, None, None] + ash.double()[:, :, None, None]
    pre_ref = F.conv2d(fa, wh.double(), bh.double(), padding=1) + x.double()
    rows = torch.full((b * h * w, cin + 8), 7.0, device="cuda", dtype=torch.bfloat16)
    rows[:, :cin] = to_rows(feat.float()).bfloat16()
    args = (rows[:, :cin], b, h, w, cin, asc.cuda(), ash.cuda(), wh.permute(0, 2, 3, 1).contiguous().cuda(), bh.cuda(), 3, x.cuda())
    o = torch.empty(b, 3, h, w, device="cuda")
    ops.head_conv3x3(*args, o, None)
    scale = float(pre_ref.abs().max())
    assert err(o, pre_ref.clamp(0, 1)) < 4e-5 * scale
    _lib.load().uwir_head_variant(0)
    try:
        o2 = torch.empty(b, 3, h, w, device="cuda")
        ops.head_conv3x3(*args, o2, None)
    finally:
        _lib.load().uwir_head_variant(1)
    assert err(o2, pre_ref.clamp(0, 1)) < 2e-5 * scale and err(o, o2) < 4e-5 * scale
    pre = torch.empty(b, 3, h, w, device="cuda")                    # a pre-clamp tap keeps the exact kernel
    ops.head_conv3x3(*args, o2, pre)
    assert err(pre, pre_ref) < 2e-5 * scale


def test_stem_mfma_into_bf16_rows(ops):
    """The student's stem into bf16 rows (bf16 mode) on the matrix cores, against fp64 and the thread-per-pixel kernel."""
    from underwater_image_restoration_amd import _lib
    g = torch.Generator().manual_seed(112)
    for cin, (b, h, w) in ((3, (2, 37, 50)), (1, (1, 8, 16)), (4, (1, 33, 17))):
        x = torch.rand(b, cin, h, w, generator=g)
        ws_, bs_ = torch.randn(64, cin, 3, 3, generator=g) * 0.2, torch.randn(64, generator=g)
        ref = F.conv2d(x.double(), ws_.double(), bs_.double(), padding=1)
        wd = ws_.permute(2, 3, 1, 0).contiguous().cuda()
        out = torch.full((b * h * w, 64 + 8), 3.0, device="cuda", dtype=torch.bfloat16)
        ops.stem_conv3x3(x.cuda(), wd, bs_.cuda(), out[:, :64])
        _lib.load().uwir_head_variant(1 | 2)
        try:
            old = torch.empty(b * h * w, 64, device="cuda", dtype=torch.bfloat16)
            ops.stem_conv3x3(x.cuda(), wd, bs_.cuda(), old)
        finally:
            _lib.load().uwir_head_variant(1)
        scale = float(ref.abs().max())
        assert err(from_rows(out[:, :64], b, h, w), ref) < 5e-3 * scale and float(out[:, 64:].float().min()) == 3.0   # bf16 rows
        # same fp32 value to ~2^-16 before the one rounding: the two bf16 results differ by at most one bf16 ulp, rarely
        d = (out[:, :64].float() - old.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * scale and float((d > 0).float().mean()) < 0.02


def test_sde_steps(ops):
    g = torch.Generator().manual_seed(12)
    for numel in (3 * 64 * 64 * 2, 1027):
        x, mu, nz, eps = (torch.randn(numel, generator=g) for _ in range(4))
        theta, sigma, sbar, dt = 0.013, 0.0316, 0.19, 0.104
        score = -nz.double() / sbar
        ref = x.double() - (theta * (mu.double() - x.double()) - sigma ** 2 * score) * dt - sigma * (eps.double() * math.sqrt(dt))
        xo = torch.empty(numel, device="cuda")
        ops.sde_euler_step(x.cuda(), mu.cuda(), nz.cuda(), eps.cuda(), xo, theta, sigma, sbar, dt)
        assert err(xo, ref) < 1e-5
        ref_ode = x.double() - (theta * (mu.double() - x.double()) - 0.5 * sigma ** 2 * score) * dt
        ops.sde_euler_step(x.cuda(), mu.cuda(), nz.cuda(), None, xo, theta, sigma, sbar, dt, 0.5)
        assert err(xo, ref_ode) < 1e-5
        expA, t1, t2, sd = 1.7, 0.8, 0.15, 0.02
        x0 = (x.double() - mu.double() - sbar * nz.double()) * expA + mu.double()
        refp = t1 * (x.double() - mu.double()) + t2 * (x0 - mu.double()) + mu.double() + sd * eps.double()
        xc = x.cuda()
        ops.sde_posterior_step(xc, mu.cuda(), nz.cuda(), eps.cuda(), xc, sbar, expA, t1, t2, sd)  # in place
        assert err(xc, refp) < 1e-5


@pytest.mark.parametrize("odt", ["fp32", "bf16"])
def test_ln_rows(ops, odt):
    g = torch.Generator().manual_seed(13)
    for (m, c, period) in ((7, 768, 0), (100, 64, 50), (5, 1028, 5), (3, 4, 0)):
        x = torch.randn(m, c + 4, generator=g) * 2 + 1
        gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
        add = torch.randn(period, c, generator=g) if period else None
        xin = x[:, :c].double() + (add.double().repeat(m // period, 1) if period else 0)
        ref = F.layer_norm(xin, (c,), gamma.double(), beta.double(), 1e-5)
        y = torch.full((m, c + 8), 3.0, device="cuda", dtype=_wdt(odt))
        ops.ln_rows(x.cuda()[:, :c], gamma.cuda(), beta.cuda(), y[:, :c], 1e-5, add=None if add is None else add.cuda(),
                    add_period=period)
        assert err(y[:, :c], ref) < (1e-5 if odt == "fp32" else 4e-2)
        assert float(y[:, c:].float().min()) == 3.0
        if odt == "fp32":  # gain-only variant (IR-SDE UNet LayerNorm) and in-place
            xc = x.cuda()
            ops.ln_rows(xc[:, :c], gamma.cuda(), None, xc[:, :c], 1e-5)
            assert err(xc[:, :c], F.layer_norm(x[:, :c].double(), (c,), gamma.double(), None, 1e-5)) < 1e-5


def test_patchify_and_add_rows(ops):
    g = torch.Generator().manual_seed(14)
    img = torch.randn(2, 3, 64, 96, generator=g)
    P = 32
    ref = F.unfold(img, kernel_size=P, stride=P).permute(0, 2, 1)  # (B, G, 3*P*P), k = (c, ky, kx)
    out = torch.full((2 * 7, 3 * P * P), 9.0, device="cuda")
    ops.patchify(img.cuda(), P, out)
    got = out.reshape(2, 7, -1).cpu()
    assert float(got[:, 0].abs().max()) == 0.0 and torch.equal(got[:, 1:], ref)
    outb = torch.empty(2 * 7, 3 * P * P, device="cuda", dtype=torch.bfloat16)
    ops.patchify(img.cuda(), P, outb)
    assert torch.equal(outb.reshape(2, 7, -1)[:, 1:].cpu(), ref.to(torch.bfloat16))
    a, b = torch.randn(9, 24, generator=g).cuda(), torch.randn(9, 24, generator=g).cuda()
    exp = a[:, :20] + b[:, 4:]
    ops.add_rows(a[:, :20], b[:, 4:], a[:, :20])
    assert torch.equal(a[:, :20], exp)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_conv_gemm_resblock_epilogue_and_upsample(ops, prec):
    """IR-SDE UNet pieces: (acc*batch_scale + batch_bias) -> SiLU with strided per-sample vectors; nearest-2x + conv3x3."""
    g = torch.Generator().manual_seed(15)
    tol = 2e-5 if prec == "fp32" else 3e-2
    b, c, h, w, n = 2, 32, 6, 10, 64
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(n, c, 3, 3, generator=g) / math.sqrt(9 * c)
    ss = torch.randn(b, 4 * n, generator=g)                     # [.. | scale+1 | shift | ..] inside one wide matrix
    wd = wt.permute(0, 2, 3, 1).reshape(n, -1).to(_wdt(prec)).cuda().contiguous()
    ssd = ss.cuda()
    out = torch.empty(b * h * w, n, device="cuda")
    ops.conv_gemm(to_rows(x), wd, out, B=b, H=h, W=w, C_=c, N=n, KH=3, KW=3, pad=1, batch_scale=ssd[:, n:2 * n],
                  batch_bias=ssd[:, 2 * n:3 * n], act=ops.ACT_SILU)
    ref = F.silu(F.conv2d(x.double(), wt.double(), padding=1) * ss[:, n:2 * n].double()[:, :, None, None]
                 + ss[:, 2 * n:3 * n].double()[:, :, None, None])
    assert rel(from_rows(out, b, h, w), ref) < tol
    bias = torch.randn(n, generator=g)
    up = torch.empty(b * 4 * h * w, n, device="cuda")
    ops.conv_gemm(to_rows(x), wd, up, B=b, H=h, W=w, C_=c, N=n, KH=3, KW=3, pad=1, ups=1, bias=bias.cuda())
    refu = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), bias.double(), padding=1)
    assert rel(from_rows(up, b, 2 * h, 2 * w), refu) < tol


@pytest.mark.parametrize("adt", ["fp32", "bf16"])
@pytest.mark.parametrize("n", [256, 192, 64])
def test_conv3x3_halo_tiles(ops, adt, n):
    """bf16 3x3 convs run over LDS-staged halo tiles: many tiles (the 128-wide channel tile needs >= 512 blocks), ragged
    image edges against the 8 x 16 pixel tile, a partial channel tile, fp32 and bf16 activations, residual epilogue."""
    g = torch.Generator().manual_seed(41)
    b, c, h, w = 2, 32, 125, 130
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(n, c, 3, 3, generator=g) / math.sqrt(9 * c)
    bias, res = torch.randn(n, generator=g), torch.randn(b, n, h, w, generator=g)
    xin = to_rows(x).to(_wdt(adt))
    xe = xin.float().cpu().reshape(b, h, w, c).permute(0, 3, 1, 2).double()
    out = torch.empty(b * h * w, n, device="cuda")
    ops.conv_gemm(xin, wt.permute(0, 2, 3, 1).reshape(n, -1).bfloat16().cuda().contiguous(), out, B=b, H=h, W=w, C_=c, N=n,
                  KH=3, KW=3, pad=1, bias=bias.cuda(), res=to_rows(res))
    ref = F.conv2d(xe, wt.bfloat16().double(), bias.double(), padding=1) + res.double()
    assert rel(from_rows(out, b, h, w), ref) < 1e-2


def test_groupnorm_affine(ops):
    g = torch.Generator().manual_seed(16)
    for (b, c, h, w) in ((2, 128, 4, 4), (1, 256, 32, 32), (3, 64, 5, 7)):
        x = torch.randn(b, c, h, w, generator=g) * 2 + torch.randn(1, c, 1, 1, generator=g) * 3
        gw, gb = torch.randn(c, generator=g), torch.randn(c, generator=g)
        ref = F.group_norm(x.double(), 32, gw.double(), gb.double(), 1e-6)
        xr = to_rows(x)
        a, s = torch.empty(b, c, device="cuda"), torch.empty(b, c, device="cuda")
        ws = torch.empty(max(ops.ln2d_workspace_bytes(b, h * w, c) // 8, 1), dtype=torch.float64, device="cuda")
        ops.groupnorm_affine(xr, b, h * w, c, 32, 1e-6, gw.cuda(), gb.cuda(), a, s, ws)
        got = xr.reshape(b, h * w, c) * a[:, None, :] + s[:, None, :]
        assert rel(from_rows(got.reshape(-1, c), b, h, w), ref) < 2e-5


@pytest.mark.parametrize("io", [("fp32", "fp32"), ("bf16", "bf16")])
def test_linear_attention(ops, io):
    g = torch.Generator().manual_seed(17)
    idt, odt = (_wdt(t) for t in io)
    for (b, n, heads) in ((2, 100, 4), (1, 1500, 4), (2, 7, 2)):
        hd = heads * 32
        qkv = (torch.randn(b * n, 3 * hd + 8, generator=g) * 2).to(idt)
        q, k, v = (qkv[:, i * hd:(i + 1) * hd].double().reshape(b, n, heads, 32).permute(0, 2, 3, 1) for i in range(3))
        qs = q.softmax(dim=-2) * 32 ** -0.5
        ks = k.softmax(dim=-1)
        ctx = torch.einsum("bhdn,bhen->bhde", ks, v / n)
        ref = torch.einsum("bhde,bhdn->bhen", ctx, qs).permute(0, 3, 1, 2).reshape(b * n, hd)
        out = torch.full((b * n, hd + 4), 2.0, device="cuda", dtype=odt)
        ws = torch.empty(ops.linattn_workspace_bytes(b, n, heads) // 4, device="cuda")
        ops.linear_attention(qkv.cuda(), b, n, heads, out[:, :hd], ws)
        scale = float(ref.abs().max())
        assert err(out[:, :hd], ref) < (2e-6 if io[0] == "fp32" else 1e-2) * max(scale, 1e-3) * 10
        assert float(out[:, hd:].float().min()) == 2.0


@pytest.mark.parametrize("cfg", [(2, 100, 64), (1, 1500, 128), (2, 64, 256), (1, 4100, 64), (8, 16384, 64)])
def test_linattn_block_fused(ops, cfg):
    """The fused Residual(PreNorm(LinearAttention)) block (csrc/linattn.hip) against an fp64 evaluation of
    module_util.py:27-33,77-97,157-185 on the same bf16-rounded weights; ragged n (not a multiple of the 32-row MFMA
    group or the 64-row tile), several chunks per batch element (online max across tiles and chunks), large |k|."""
    b, n, dim = cfg
    g = torch.Generator().manual_seed(31)
    ld = dim + 4
    x = torch.randn(b * n, ld, generator=g) * 1.5 + 0.3
    x[: n // 2, :dim] *= 3.0                                           # k grows along the rows: the running max must move
    g_pre, g_out = torch.rand(dim, generator=g) + 0.5, torch.rand(dim, generator=g) + 0.5
    w_qkv = (torch.randn(384, dim, generator=g) * (2.0 / dim ** 0.5)).bfloat16()
    w_out = (torch.randn(dim, 128, generator=g) * (1.0 / 128 ** 0.5)).bfloat16()
    b_out = torch.randn(dim, generator=g) * 0.1

    def chan_ln(t, gain):                                               # (rows, C): module_util.py:77-86
        mu = t.mean(dim=1, keepdim=True)
        var = ((t - mu) ** 2).mean(dim=1, keepdim=True)
        return (t - mu) / torch.sqrt(var + 1e-5) * gain

    xd = x[:, :dim].double().cuda()
    xn = chan_ln(xd, g_pre.double().cuda())
    qkv = xn @ w_qkv.double().cuda().T
    q, k, v = (qkv[:, i * 128:(i + 1) * 128].reshape(b, n, 4, 32).permute(0, 2, 3, 1) for i in range(3))
    ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v / n)
    la = torch.einsum("bhde,bhdn->bhen", ctx, q.softmax(dim=-2) * 32 ** -0.5).permute(0, 3, 1, 2).reshape(b * n, 128)
    ref = chan_ln(la @ w_out.double().cuda().T + b_out.double().cuda(), g_out.double().cuda()) + xd
    out = torch.full((b * n, dim + 8), 7.0, device="cuda")
    ws = torch.empty(ops.linattn_block_workspace_bytes(b, n) // 4, device="cuda")
    ops.linattn_block(x.cuda()[:, :dim], b, n, dim, g_pre.cuda(), w_qkv.cuda(), w_out.cuda(), b_out.cuda(), g_out.cuda(),
                      out[:, :dim], ws)
    e = err(out[:, :dim], ref)
    # the LayerNorm'ed attention term has unit scale: bf16 operands (xn, E, V, P, ctx, Y) give ~1e-2 absolute
    print(f"\n[linattn_block b{b} n{n} dim{dim}] max abs err {e:.3e} (|ref|max {float(ref.abs().max()):.2f})")
    assert e < 6e-2 and float(out[:, dim:].min()) == 7.0
    rel_l2 = float((out[:, :dim].double() - ref).norm() / (ref - xd).norm())
    assert rel_l2 < 1.5e-2


def test_geglu_softmax_input_tail(ops):
    g = torch.Generator().manual_seed(18)
    h = torch.randn(37, 2 * 64, generator=g)
    for dt in (torch.float32, torch.bfloat16):
        o = torch.empty(37, 64, device="cuda", dtype=dt)
        ops.geglu(h.to(dt).cuda(), o)
        hh = h.to(dt).double()
        assert err(o, hh[:, :64] * F.gelu(hh[:, 64:])) < (1e-5 if dt == torch.float32 else 3e-2)
    x, mul = torch.randn(5, 256, generator=g) * 3, torch.rand(256, generator=g)
    y = torch.zeros(5, 300, device="cuda")
    ops.softmax_rows(x.cuda(), mul.cuda(), y[:, 20:276])
    assert err(y[:, 20:276], torch.softmax(x.double(), dim=1) * mul.double()) < 1e-6
    # input assembly: cat([xt - cond, cond]) + reflect pad, NHWC with zero-filled channels
    xt, cond = torch.randn(2, 3, 13, 18, generator=g), torch.randn(2, 3, 13, 18, generator=g)
    Hp, Wp = 16, 32
    ref = F.pad(torch.cat([xt - cond, cond], dim=1), (0, Wp - 18, 0, Hp - 13), "reflect")
    out = torch.empty(2 * Hp * Wp, 8, device="cuda")
    ops.unet_input(xt.cuda(), cond.cuda(), Hp, Wp, out)
    got = out.reshape(2, Hp, Wp, 8).permute(0, 3, 1, 2).cpu()
    assert torch.equal(got[:, :6], ref) and float(got[:, 6:].abs().max()) == 0.0
    # tail conv3x3 + crop, NCHW output
    feat = torch.randn(2, 32, Hp, Wp, generator=g)
    wt, bt = torch.randn(3, 32, 3, 3, generator=g) * 0.1, torch.randn(3, generator=g)
    o = torch.empty(2, 3, 13, 18, device="cuda")
    ops.tail_conv3x3(to_rows(feat), 2, Hp, Wp, 32, wt.permute(0, 2, 3, 1).contiguous().cuda(), bt.cuda(), 3, o)
    assert err(o, F.conv2d(feat.double(), wt.double(), bt.double(), padding=1)[..., :13, :18]) < 2e-5


@pytest.mark.parametrize("cfg", [(2, 200, 3, 16, 1), (1, 333, 4, 16, 2), (1, 130, 2, 32, 1), (1, 97, 2, 64, 2), (1, 4096, 2, 16, 1)])
def test_flash_attn_bf16_fast_path_and_fallback(ops, cfg):
    """bf16 fast path (a-priori bound |q| max|k| as the softmax shift) vs the online-max loop, and the automatic
    fallback when the bound is too large for a fixed shift (scores of several hundred log2 units)."""
    b, n, heads, hd, nv = cfg
    g = torch.Generator().manual_seed(19)
    wdt = heads * hd
    for amp, expect_fast in ((1.0, True), (12.0, False)):
        q, k = (torch.randn(b * n, wdt, generator=g) * amp).bfloat16(), (torch.randn(b * n, wdt, generator=g) * amp).bfloat16()
        vs = [torch.randn(b * n, wdt, generator=g).bfloat16() for _ in range(nv)]
        refs = _attn_ref(q.double().reshape(b, n, wdt), k.double().reshape(b, n, wdt), [t.double().reshape(b, n, wdt) for t in vs],
                         heads, hd)
        outs = {}
        for fast in (True, False):
            o = [torch.empty(b * n, wdt, device="cuda") for _ in range(nv)]
            ops.flash_attn(q.cuda(), k.cuda(), vs[0].cuda(), vs[1].cuda() if nv == 2 else None, b, n, heads, hd,
                           1.0 / math.sqrt(hd), o[0], o[1] if nv == 2 else None, fast=fast)
            outs[fast] = o
            if expect_fast:  # (at amp 12 the softmax is near one-hot and bf16 score rounding alone moves O(1) mass)
                for got, ref in zip(o, refs):
                    assert err(got.reshape(b, n, wdt), ref) < 3e-2, (amp, fast)
        if not expect_fast:  # bound > 60 everywhere: the fast call must have taken the very same online-max code
            for a_, b_ in zip(outs[True], outs[False]):
                assert torch.equal(a_, b_)


def _attn_ref_cuda(q, k, vs, b, n, heads, hd):
    """fp64 reference on the device, one (batch, head) at a time (long sequences: the n x n matrix stays small)."""
    outs = [torch.empty(b * n, heads * hd, dtype=torch.float64, device="cuda") for _ in vs]
    for bi in range(b):
        rows = slice(bi * n, (bi + 1) * n)
        for h in range(heads):
            cols = slice(h * hd, (h + 1) * hd)
            a = torch.softmax(q[rows, cols].double() @ k[rows, cols].double().T / math.sqrt(hd), dim=-1)
            for o, v in zip(outs, vs):
                o[rows, cols] = a @ v[rows, cols].double()
    return outs


@pytest.mark.parametrize("cfg", [(1, 4096, 16, 16, 1), (1, 2100, 32, 16, 2), (1, 2049, 32, 32, 1), (2, 2500, 13, 32, 2),
                                 (1, 2048, 32, 16, 1), (2, 1030, 64, 32, 1), (8, 1024, 8, 32, 1)])
def test_flash_attn_packed_stream(ops, cfg):
    """Long bf16 sequences with hd 16 / 32 take the packed-stream kernels (pre-packed K / V^T fragments, no LDS):
    parity with the fp64 softmax attention, agreement with the LDS-staged kernels, ragged n (keys padded to 32,
    query blocks partly empty), strided q|k|v views, and the per-block fallback for unbounded scores."""
    b, n, heads, hd, nv = cfg
    lib_need = ops._lib.load().uwir_flash_attn_packed_workspace_bytes(b, n, heads, hd, nv)
    assert lib_need > ops._lib.load().uwir_flash_attn_workspace_bytes(b, n, heads) + n * heads * hd  # the packed path is on
    g = torch.Generator().manual_seed(23)
    wdt = heads * hd
    qkv = (torch.randn(b * n, 3 * wdt + 8, generator=g)).bfloat16().cuda()
    q, k, v = qkv[:, :wdt], qkv[:, wdt:2 * wdt], qkv[:, 2 * wdt:3 * wdt]
    v2 = torch.randn(b * n, wdt, generator=g).bfloat16().cuda() if nv == 2 else None
    vs = [v] + ([v2] if nv == 2 else [])
    refs = _attn_ref_cuda(q, k, vs, b, n, heads, hd)
    outs = {}
    for packed in (True, False):
        ops._PACKED = packed
        try:
            o = [torch.full((b * n, wdt + 4), -9.0, device="cuda") for _ in range(nv)]
            ops.flash_attn(q, k, v, v2, b, n, heads, hd, 1.0 / math.sqrt(hd), o[0][:, :wdt], o[1][:, :wdt] if nv == 2 else None)
        finally:
            ops._PACKED = True
        outs[packed] = o
        for got, ref in zip(o, refs):
            assert err(got[:, :wdt], ref) < 3e-2, packed
            assert float(got[:, wdt:].max()) == -9.0
    for a_, b_ in zip(outs[True], outs[False]):  # same bf16 operands, same exp2: only the summation order differs
        assert float((a_ - b_).abs().max()) < 2e-3
    # one batch element with huge scores: its blocks must fall back to the online-max loop (bit-equal to it), the
    # others stay on the stream kernel
    q2 = q.clone()
    q2[:n] = (q2[:n].float() * 40).bfloat16()
    o_p = torch.empty(b * n, wdt, device="cuda")
    o_o = torch.empty(b * n, wdt, device="cuda")
    ops.flash_attn(q2, k, v, None, b, n, heads, hd, 1.0 / math.sqrt(hd), o_p, None)
    ops.flash_attn(q2, k, v, None, b, n, heads, hd, 1.0 / math.sqrt(hd), o_o, None, fast=False)
    assert torch.equal(o_p[:n], o_o[:n]) and bool(torch.isfinite(o_p).all())
    if b > 1:
        assert err(o_p[n:], refs[0][n:]) < 3e-2


@pytest.mark.parametrize("cfg", [(2, 4096 + 7, 8, 16, 2), (1, 2048 + 19, 32, 32, 1)])
def test_flash_attn_per_dimension_bound(ops, cfg):
    """Packed-stream attention: a few keys with a large component in a dimension the queries do not use push the
    Cauchy-Schwarz bound |c q| max|k| far above 90 while every score stays small.  The per-dimension bound (k_d in
    [kmin_d, kmax_d], extremes from the pack pre-pass; keys padded to 32 count as zeros) keeps those queries on the
    no-shift stream kernel; queries that DO use the dimension still fall back.  One dimension holds positive keys only."""
    from underwater_image_restoration_amd import _lib
    lib = _lib.load()
    b, n, heads, hd, nv = cfg
    assert lib.uwir_flash_attn_packed_workspace_bytes(b, n, heads, hd, nv) > lib.uwir_flash_attn_workspace_bytes(b, n, heads) + n * heads * hd  # the packed path is on
    g = torch.Generator().manual_seed(31)
    wdt = heads * hd
    c = 1.4426950408889634 / math.sqrt(hd)
    q = torch.randn(b * n, heads, hd, generator=g)
    k = torch.randn(b * n, heads, hd, generator=g)
    k[:, :, 1] = k[:, :, 1].abs() + 1.0                     # kmin_1 > 0 among the real keys
    q[:, :, 0] = 0.0
    out_rows = torch.randperm(b * n, generator=g)[:24]
    k[out_rows, :, 0] = 300.0                               # outlier keys, positive only: kmin_0 stays near -4
    hot = torch.arange(0, 256)                              # the first 256 queries of batch 0 use dimension 0: fallback
    q[hot, :, 0] = 6.0
    q, k = q.reshape(b * n, wdt).bfloat16().cuda(), k.reshape(b * n, wdt).bfloat16().cuda()
    v = torch.randn(b * n, wdt, generator=g).bfloat16().cuda()
    v2 = torch.randn(b * n, wdt, generator=g).bfloat16().cuda() if nv == 2 else None
    vs = [v] + ([v2] if nv == 2 else [])
    qn = q.float().reshape(b * n, heads, hd).norm(dim=-1) * c
    assert float(qn.min()) * 300.0 > 90.0                   # the norm bound rejects every query
    refs = _attn_ref_cuda(q, k, vs, b, n, heads, hd)
    outs = {}
    for variant in (8, 40, "online"):                       # 8: default forms; 40: norm bound alone
        if variant != "online":
            lib.uwir_attn_variant(variant)
        try:
            o = [torch.empty(b * n, wdt, device="cuda") for _ in range(nv)]
            ops.flash_attn(q, k, v, v2, b, n, heads, hd, 1.0 / math.sqrt(hd), o[0], o[1] if nv == 2 else None,
                           fast=variant != "online")
            torch.cuda.synchronize()
        finally:
            lib.uwir_attn_variant(-1)
        outs[variant] = o
        for got, ref in zip(o, refs):
            assert bool(torch.isfinite(got).all())
            assert err(got, ref) < 3e-2, variant
    for t, w, o in zip(outs[8], outs[40], outs["online"]):
        assert torch.equal(w, o)                            # norm bound alone: everything fell back to the online-max loop
        assert torch.equal(t[:256], o[:256])                # the hot queries still do
        assert not torch.equal(t[256:], o[256:])            # the rest ran the stream kernel


@pytest.mark.parametrize("cfg", [(1, 2048 + 19, 32, 16, 2), (1, 2048 + 5, 32, 32, 1), (1, 4096 + 30, 16, 16, 1)])
def test_flash_attn_ragged_tail_with_very_negative_scores(ops, cfg):
    """Packed-stream attention on a sequence that is not a multiple of 32 keys while EVERY score of every query is about
    -50 (log2 units, inside the no-shift bound): the zero-padded keys must not reach the row sums (they used to be counted
    as exp2(0) = 1 each and subtracted afterwards, which cancelled to 0 / 0 here)."""
    b, n, heads, hd, nv = cfg
    g = torch.Generator().manual_seed(5)
    wdt = heads * hd
    amp = 12.0 * (hd / 16) ** 0.25                          # q . k / sqrt(hd) * log2(e) stays near -52
    u = torch.nn.functional.normalize(torch.randn(heads, hd, generator=g), dim=-1)
    q = (u[None] * amp + 0.1 * torch.randn(n, heads, hd, generator=g)).reshape(n, wdt).bfloat16().cuda()
    k = (-u[None] * amp + 0.3 * torch.randn(n, heads, hd, generator=g)).reshape(n, wdt).bfloat16().cuda()
    v = torch.randn(n, wdt, generator=g).bfloat16().cuda()
    v2 = torch.randn(n, wdt, generator=g).bfloat16().cuda() if nv == 2 else None
    vs = [v] + ([v2] if nv == 2 else [])
    refs = _attn_ref_cuda(q, k, vs, b, n, heads, hd)
    o = [torch.empty(n, wdt, device="cuda") for _ in range(nv)]
    ops.flash_attn(q, k, v, v2, b, n, heads, hd, 1.0 / math.sqrt(hd), o[0], o[1] if nv == 2 else None)
    r = [torch.empty(n, wdt, device="cuda") for _ in range(nv)]
    ops.flash_attn(q, k, v, v2, b, n, heads, hd, 1.0 / math.sqrt(hd), r[0], r[1] if nv == 2 else None, fast=False)
    for got, onl, ref in zip(o, r, refs):
        assert bool(torch.isfinite(got).all())
        assert err(got, ref) < 3e-2 and err(onl, ref) < 3e-2
    assert not torch.equal(o[0], r[0])                      # the stream kernel, not the online-max fallback, produced o


def test_flash_attn_bound_band_60_to_90(ops):
    """The no-shift fast path is taken while the a-priori score bound ub = |c q| max|k| is <= 90 (log2 units).  Drive the
    (60, 90] band on the packed-stream kernel at n = 65536: every query has ub ~ 85 and keys aligned with / against the
    common query direction give scores from about -85 to +85, i.e. P up to 2^85 and row sums up to n * 2^85 in fp32."""
    b, n, heads, hd = 1, 65536, 2, 16
    g = torch.Generator().manual_seed(29)
    c = 1.4426950408889634 / math.sqrt(hd)
    u = torch.nn.functional.normalize(torch.randn(heads, hd, generator=g), dim=-1)
    qd = torch.nn.functional.normalize(u[None] + 0.05 * torch.randn(n, heads, hd, generator=g), dim=-1)
    q = (qd * (8.5 / c)).reshape(n, heads * hd)                        # |c q| = 8.5
    sgn = torch.where(torch.rand(n, heads, 1, generator=g) < 0.5, -1.0, 1.0)
    kd = torch.nn.functional.normalize(u[None] * sgn + 0.3 * torch.randn(n, heads, hd, generator=g), dim=-1)
    k = (kd * (10.0 * torch.rand(n, heads, 1, generator=g))).reshape(n, heads * hd)
    k[7] = (u * 10.0).reshape(-1)                                      # max|k| = 10 exactly -> ub = 85 (+ margin)
    v = torch.randn(n, heads * hd, generator=g)
    q, k, v = q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda()
    ub = float((q.float().reshape(n, heads, hd).norm(dim=-1) * c).max() * k.float().reshape(n, heads, hd).norm(dim=-1).max())
    assert 60.0 < ub * 1.002 + 1e-3 <= 90.0
    ref = torch.empty(n, heads * hd, dtype=torch.float64, device="cuda")
    smax = 0.0
    for h in range(heads):
        cols = slice(h * hd, (h + 1) * hd)
        kk, vv = k[:, cols].double(), v[:, cols].double()
        for r0 in range(0, n, 4096):
            sc = q[r0:r0 + 4096, cols].double() @ kk.T / math.sqrt(hd)
            smax = max(smax, float(sc.abs().max()) * 1.4426950408889634)
            ref[r0:r0 + 4096, cols] = torch.softmax(sc, dim=-1) @ vv
    assert smax > 60.0                                                 # scores really reach the band (log2 units)
    o_f = torch.empty(n, heads * hd, device="cuda")
    o_o = torch.empty(n, heads * hd, device="cuda")
    ops.flash_attn(q, k, v, None, b, n, heads, hd, 1.0 / math.sqrt(hd), o_f, None)
    ops.flash_attn(q, k, v, None, b, n, heads, hd, 1.0 / math.sqrt(hd), o_o, None, fast=False)
    assert bool(torch.isfinite(o_f).all())
    assert err(o_f, ref) < 3e-2 and err(o_o, ref) < 3e-2
    assert not torch.equal(o_f, o_o)                                   # the two launches are different kernels


def test_flash_attn_two_v_fallback_at_65536(ops):
    """configs[4]'s CFC attention (cfc.py:24-39) at its production length: two V's, head_dim 16, 4 heads, n = 65536 (a 512 x 512
    frame at level 0).  Outlier keys (one dimension at 300) defeat the norm bound for every query; the per-dimension bound
    keeps the queries that do not use that dimension on the no-shift stream kernel, and every query of each FOURTH
    128-query group does use it: exactly 25 % of the groups (128 of 512 per head) must fall back to the online-max
    kernel -- bit-equal to the fast=False launch there, different from it elsewhere -- and all of it must match the
    chunked fp64 softmax attention."""
    from underwater_image_restoration_amd import _lib
    lib = _lib.load()
    b, n, heads, hd = 1, 65536, 4, 16
    assert lib.uwir_flash_attn_packed_workspace_bytes(b, n, heads, hd, 2) > lib.uwir_flash_attn_workspace_bytes(b, n, heads) + n * heads * hd
    g = torch.Generator().manual_seed(65)
    wdt = heads * hd
    c = 1.4426950408889634 / math.sqrt(hd)
    q = torch.randn(n, heads, hd, generator=g)
    k = torch.randn(n, heads, hd, generator=g)
    q[:, :, 0] = 0.0
    k[torch.randperm(n, generator=g)[:40], :, 0] = 300.0      # outlier keys
    hot = (torch.arange(n) // 128) % 4 == 1                    # every fourth 128-query group
    q[hot, :, 0] = 6.0                                         # scores up to 6 * 300 / 4 * log2(e) ~ 650 there
    q, k = q.reshape(n, wdt).bfloat16().cuda(), k.reshape(n, wdt).bfloat16().cuda()
    vt, vf = torch.randn(n, wdt, generator=g).bfloat16().cuda(), torch.randn(n, wdt, generator=g).bfloat16().cuda()
    assert float((q.float().reshape(n, heads, hd).norm(dim=-1) * c).min()) * 300.0 > 90.0   # the norm bound rejects everything
    refs = [torch.empty(n, wdt, dtype=torch.float64, device="cuda") for _ in range(2)]
    for h in range(heads):
        cols = slice(h * hd, (h + 1) * hd)
        kk = k[:, cols].double()
        for r0 in range(0, n, 2048):
            a = torch.softmax(q[r0:r0 + 2048, cols].double() @ kk.T / math.sqrt(hd), dim=-1)
            refs[0][r0:r0 + 2048, cols] = a @ vt[:, cols].double()
            refs[1][r0:r0 + 2048, cols] = a @ vf[:, cols].double()
    o = [torch.empty(n, wdt, device="cuda") for _ in range(2)]
    r = [torch.empty(n, wdt, device="cuda") for _ in range(2)]
    ops.flash_attn(q, k, vt, vf, b, n, heads, hd, 1.0 / math.sqrt(hd), o[0], o[1])
    ops.flash_attn(q, k, vt, vf, b, n, heads, hd, 1.0 / math.sqrt(hd), r[0], r[1], fast=False)
    torch.cuda.synchronize()
    hot = hot.cuda()
    for got, onl, ref in zip(o, r, refs):
        assert bool(torch.isfinite(got).all())
        assert err(got, ref) < 3e-2 and err(onl, ref) < 3e-2
        assert torch.equal(got[hot], onl[hot])                 # the marked groups were redone by the online-max kernel
        same_rows = (got[~hot] == onl[~hot]).all(dim=1).float().mean()
        assert float(same_rows) < 0.01                         # the other 75 % came from the stream kernel


def test_affine_cast(ops):
    g = torch.Generator().manual_seed(20)
    b, n, c = 3, 37, 24
    x, a, s = torch.randn(b * n, c + 4, generator=g), torch.randn(b, c, generator=g), torch.randn(b, c, generator=g)
    ref = x[:, :c].reshape(b, n, c).double() * a.double()[:, None] + s.double()[:, None]
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2)):
        out = torch.full((b * n, c + 8), 4.0, device="cuda", dtype=dt)
        ops.affine_cast(x.cuda()[:, :c], b, n, c, a.cuda(), s.cuda(), out[:, :c])
        assert err(out[:, :c].reshape(b, n, c), ref) < tol and float(out[:, c:].float().min()) == 4.0


@pytest.mark.parametrize("shape", [(4096 + 77, 192, 1024), (8192, 576, 1728), (4500, 384, 2048), (4096 + 77, 288, 1088),
                                   (4608, 864, 1024)])
@pytest.mark.parametrize("epi", ["plain", "bias_scale", "gelu"])
def test_conv_gemm_wide_tiles(ops, shape, epi):
    """Large-K bf16 1x1 GEMMs take the 256 x 288 kernel when N % 288 == 0 (csrc/gemm_wide288.hip) and the 256 x 192 ring
    kernel when N % 192 == 0 (csrc/gemm_wide.hip): ragged M, an odd number of K stages, strided rows, every epilogue they
    accept, against an fp64 reference, against the 128 x 128 kernel (same sums in another order) and -- where both wide
    kernels apply -- against each other (the same sums in the same order: bit-equal)."""
    from underwater_image_restoration_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    Abuf = torch.randn(M, K + 8, generator=g).cuda().bfloat16()
    A = Abuf[:, :K]                                             # lda = K + 8
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda() if epi != "plain" else None
    cs = (torch.rand(N, generator=g) + 0.5).cuda() if epi == "bias_scale" else None
    act = ops.ACT_GELU if epi == "gelu" else ops.ACT_NONE
    outs = []
    for variant in (2, 64, 0):   # 2: keep the 128 x 128 kernel; 64: not the 256 x 288 one
        lib.uwir_conv_variant(variant)
        try:
            obuf = torch.full((M, N + 16), float("nan"), device="cuda", dtype=torch.bfloat16)
            out = obuf[:, 8:8 + N]                              # ldo = N + 16, 16-byte aligned
            ops.conv_gemm(A, W, out, B=1, H=1, W=M, C_=K, N=N, bias=bias, col_scale=cs, act=act)
            torch.cuda.synchronize()
        finally:
            lib.uwir_conv_variant(0)
        assert bool(torch.isnan(obuf[:, :8]).all()) and bool(torch.isnan(obuf[:, 8 + N:]).all())   # nothing outside the view
        outs.append(out.float())
    ref = A.double() @ W.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if cs is not None:
        ref = ref * cs.double()
    for o in outs:
        assert bool(torch.isfinite(o).all())
        assert rel(o, ref.float()) < 4e-3          # bf16 output rounding
    assert float((outs[0] - outs[2]).abs().max()) <= 2 ** -6 * float(ref.abs().max())   # at most one bf16 ulp apart
    if N % 288 == 0 and N % 192 == 0:
        assert torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("case", [(64, 16, 2, 16, 24), (128, 0, 3, 6, 10), (256, 16, 1, 8, 8)])
def test_gate_idwt_fused_against_the_three_launches(ops, case):
    """WFIGate + idwt2 in one launch (csrc/gate.hip) against conv_gemm (relu) -> conv_gemm (sigmoid, blend) -> idwt2 on the
    same operands, and against fp64 on the bf16-rounded operands: ragged row counts (M % 16 != 0, tiles that straddle two
    images), with and without mask columns, strided hf / out rows."""
    C, Mp, B, h, w = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(C + Mp)
    M, Kg = B * h * w, 2 * C + Mp
    gate_in = torch.randn(M, Kg, generator=g)
    if Mp:
        gate_in[:, 2 * C + 10:] = 0.0                              # the zero padding behind the 10 mask columns
    w1 = (torch.randn(64, Kg, generator=g) / Kg ** 0.5).to(bf)
    w2 = (torch.randn(C, 64, generator=g) / 8).to(bf)
    bias1, b2 = torch.randn(B, 64, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    hfb = torch.randn(M, 3 * C + 8, generator=g).to(bf)
    gi, w1c, w2c, bb, b2c, hfc = gate_in.cuda(), w1.cuda(), w2.cuda(), bias1.cuda(), b2.cuda(), hfb.cuda()
    hf = hfc[:, :3 * C]
    # the three launches
    ghid = torch.empty(M, 64, device="cuda", dtype=bf)
    ops.conv_gemm(gi, w1c, ghid, B=B, H=h, W=w, C_=Kg, N=64, batch_bias=bb, act=ops.ACT_RELU)
    alpha0, merged0 = torch.empty(B, h, w, C, device="cuda"), torch.empty(M, C, device="cuda")
    ops.conv_gemm(ghid, w2c, alpha0.view(M, C), B=B, H=h, W=w, C_=64, N=C, bias=b2c, act=ops.ACT_SIGMOID,
                  blend_t=gi[:, :C], blend_f=gi[:, C:2 * C], blend_out=merged0)
    out0 = torch.full((4 * M, C + 8), 7.0, device="cuda", dtype=bf)
    ops.idwt2(merged0, hf, B, h, w, C, out0[:, :C])
    # one launch
    alpha1, merged1 = torch.empty(B, h, w, C, device="cuda"), torch.empty(M, C, device="cuda")
    out1 = torch.full((4 * M, C + 8), 7.0, device="cuda", dtype=bf)
    ops.gate_idwt(gi, w1c, w2c, bb, b2c, hf, B, h, w, C, out1[:, :C], alpha1, merged1)
    torch.cuda.synchronize()
    assert float(out1[:, C:].float().min()) == 7.0 and float(out1[:, C:].float().max()) == 7.0     # nothing outside the view
    # fp64 on the same bf16-rounded operands (the hidden rows rounded to bf16 as both forms do)
    hid = torch.relu(gate_in.to(bf).double() @ w1.double().t() + bias1.double().repeat_interleave(h * w, 0)).to(bf).double()
    a_ref = torch.sigmoid(hid @ w2.double().t() + b2.double())
    m_ref = a_ref * gate_in[:, :C].double() + (1 - a_ref) * gate_in[:, C:2 * C].double()
    assert err(alpha1.view(M, C), a_ref) < 3e-3 and err(merged1, m_ref) < 1e-2       # bf16 rounding flips of the hidden rows
    assert err(alpha1, alpha0.double()) < 3e-3 and err(merged1, merged0.double()) < 1e-2
    # inverse transform of ITS OWN merged rows: exact up to the output's bf16 rounding
    k = float(torch.tensor(0.70710678118654752440, dtype=torch.float32) ** 2)
    mm = merged1.double().view(B, h, w, C)
    hh_ = hf.double().view(B, h, w, 3, C)
    lh, hl, hhh = hh_[..., 0, :], hh_[..., 1, :], hh_[..., 2, :]
    ref = torch.empty(B, 2 * h, 2 * w, C, dtype=torch.float64, device="cuda")
    ref[:, 0::2, 0::2] = k * (mm + lh + hl + hhh)
    ref[:, 0::2, 1::2] = k * (mm - lh + hl - hhh)
    ref[:, 1::2, 0::2] = k * (mm + lh - hl - hhh)
    ref[:, 1::2, 1::2] = k * (mm - lh - hl + hhh)
    got = out1[:, :C].double().view(B, 2 * h, 2 * w, C)
    assert float(((got - ref).abs() / (ref.abs() + 1.0)).max()) < 2 ** -8
    assert err(out1[:, :C], out0[:, :C].double()) < 5e-2
    # without the diagnostic output
    alpha2 = torch.empty_like(alpha1)
    out2 = torch.full_like(out1, 7.0)
    ops.gate_idwt(gi, w1c, w2c, bb, b2c, hf, B, h, w, C, out2[:, :C], alpha2, None)
    assert torch.equal(out2, out1) and torch.equal(alpha2, alpha1)
    assert not ops.gate_idwt_supported(48, 96 + 16) and ops.gate_idwt_supported(64, 128)
    # blend-only form (uwir_gate_blend: a block followed by another block of its stage): the same alpha, and merged as the
    # bf16 rounding of the very values the fused form blends -- bit for bit, strided destination
    alpha3 = torch.empty_like(alpha1)
    m16 = torch.full((M, C + 8), 7.0, device="cuda", dtype=bf)
    ops.gate_blend(gi, w1c, w2c, bb, b2c, B, h, w, C, alpha3, m16[:, :C])
    assert torch.equal(alpha3, alpha1) and torch.equal(m16[:, :C], merged1.to(bf))
    assert float(m16[:, C:].float().min()) == 7.0 and float(m16[:, C:].float().max()) == 7.0


@pytest.mark.parametrize("case", [(192, 3, 512), (384, 2, 384)])
def test_ffn_fused_against_the_two_launches(ops, case):
    """ffn.0 -> GELU -> ffn.2 + residual in one launch (csrc/ffn.hip) against the two uwir_conv_gemm launches it replaces
    (per-image folded weights, batched bias) and against fp64 on the same bf16 operands; strided rows, several passes per
    block, images whose weights differ."""
    C, B, n = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(C)
    M = B * n
    xb = torch.randn(M, C + 8, generator=g).to(bf).cuda()
    x = xb[:, :C]
    w0b = (torch.randn(B, 2 * C, C, generator=g) / C ** 0.5).to(bf).cuda()
    bias0 = (torch.randn(B, 2 * C, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, 2 * C, generator=g) / (2 * C) ** 0.5).to(bf).cuda()
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    assert ops.ffn_fused_supported(C, n) and not ops.ffn_fused_supported(C, n + 16) and not ops.ffn_fused_supported(768, 1024)
    # two launches
    hidden = torch.empty(M, 2 * C, device="cuda", dtype=bf)
    ops.conv_gemm(x, w0b, hidden, B=B, H=1, W=n, C_=C, N=2 * C, batch_bias=bias0, act=ops.ACT_GELU, w_batched=True)
    out0 = torch.empty(M, C, device="cuda", dtype=bf)
    ops.conv_gemm(hidden, w2, out0, B=B, H=1, W=n, C_=2 * C, N=C, bias=b2, res=x)
    # one launch
    w2p = ops.ffn_permute_w2(w2)
    obuf = torch.full((M, C + 16), 5.0, device="cuda", dtype=bf)
    out1 = obuf[:, 8:8 + C]
    ops.ffn_fused(x, w0b, bias0, w2p, b2, B, n, C, out1)
    torch.cuda.synchronize()
    assert float(obuf[:, :8].float().min()) == 5.0 and float(obuf[:, 8 + C:].float().max()) == 5.0
    # fp64 with the hidden rows rounded to bf16 as both forms do
    xd = x.double().view(B, n, C)
    hid = torch.einsum("bnc,bhc->bnh", xd, w0b.double()) + bias0.double()[:, None]
    hid = torch.nn.functional.gelu(hid).to(bf).double()
    ref = (hid.view(M, 2 * C) @ w2.double().t() + b2.double() + x.double())
    assert rel(out1, ref) < 6e-3 and rel(out0, ref) < 6e-3          # bf16 output rounding + rounding flips of hidden rows
    d = (out1.float() - out0.float()).abs()
    assert float(d.max()) <= 2 ** -6 * float(ref.abs().max())        # at most about one bf16 ulp apart
    assert float((d > 0).float().mean()) < 0.05


def test_gemm_wide_probe_matches_conv_gemm_and_rejects_other_shapes(ops):
    """uwir_gemm_wide_probe(variant 0) is the kernel conv_gemm launches for the merge shapes; shapes outside the 256 x 288
    kernel's range and unknown variants are errors."""
    from underwater_image_restoration_amd import _lib
    lib = _lib.load()
    M, N, K = 4096, 576, 1728
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    o1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o2 = torch.empty_like(o1)
    ops.conv_gemm(A, W, o1, B=1, H=1, W=M, C_=K, N=N, bias=bias)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.uwir_gemm_wide_probe(A.data_ptr(), W.data_ptr(), o2.data_ptr(), bias.data_ptr(), M, N, K, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    for v in (1, 2, 3, 4, 5):                                   # the ablations run to completion (their output is undefined)
        assert lib.uwir_gemm_wide_probe(A.data_ptr(), W.data_ptr(), o2.data_ptr(), bias.data_ptr(), M, N, K, v, st) == 0
    torch.cuda.synchronize()
    assert lib.uwir_gemm_wide_probe(A.data_ptr(), W.data_ptr(), o2.data_ptr(), bias.data_ptr(), M, N, K, 6, st) != 0
    assert lib.uwir_gemm_wide_probe(A.data_ptr(), W.data_ptr(), o2.data_ptr(), None, M, 192, K, 0, st) != 0
    assert b"288" in lib.uwir_last_error()


def test_operand_on_another_device_is_an_error(ops):
    """the library launches in the CURRENT HIP device context, so an operand that lives elsewhere must be
    refused, not enqueued on the wrong GPU's stream.  (The model-level forwards switch to their input's device
    themselves: second test, needs two GPUs.)"""
    cur = torch.cuda.current_device()
    with pytest.raises(ValueError, match="current HIP device"):
        ops._stream(torch.device("cuda", cur + 1))
    with pytest.raises(ValueError, match="current HIP device"):
        ops._stream(torch.zeros(1))
    assert ops._stream(torch.zeros(1, device="cuda")) == torch.cuda.current_stream().cuda_stream


def test_model_on_non_current_device(ops):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
    mk = dict(base_channels=16, num_levels=2, blocks_per_level=1, cond_dim=12, num_masks=0, num_degradation_types=3)
    sd = synth_state_dict(student_param_shapes(**mk), seed=3)
    x, z, _ = synth_student_inputs(1, 32, 32, 12, 0, seed=4)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        model = NAFNetWFIGate(**mk, precision="fp32")
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        assert torch.cuda.current_device() == 0
        outs.append(model(x.to(dev), z.to(dev), None)[0].cpu())   # cuda:1 is NOT the current device here
    assert torch.equal(outs[0], outs[1])


def test_unet_and_sampler_on_non_current_device(ops):
    """ConditionalUNet.forward lost its device decorators once; a UNet (and an IRSDE loop around it) on a
    GPU that is not the current one must run there and give the same result as on the current one."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from underwater_image_restoration_amd.sde import IRSDE
    from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes
    from underwater_image_restoration_amd.unet import ConditionalUNet
    mk = dict(in_nc=3, out_nc=3, nf=32, ch_mult=[1, 2], context_dim=64, use_degra_context=True, use_image_context=True)
    sd = synth_state_dict(unet_param_shapes(**mk), seed=5)
    g = torch.Generator().manual_seed(6)
    mu = torch.rand(1, 3, 32, 32, generator=g)
    x0 = mu + torch.randn(mu.shape, generator=g) * 0.1
    tc, ic = torch.randn(1, 64, generator=g), torch.randn(1, 64, generator=g)
    tape = torch.randn(3, 1, 3, 32, 32, generator=g)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        net = ConditionalUNet(mk["in_nc"], mk["out_nc"], mk["nf"], mk["ch_mult"], mk["context_dim"], True, True, precision="fp32")
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        assert torch.cuda.current_device() == 0
        eps = net(x0.to(dev), mu.to(dev), 3, text_context=tc.to(dev), image_context=ic.to(dev))
        assert eps.device == torch.device(dev) and not eps.requires_grad
        sde = IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device=dev)
        sde.set_mu(mu.to(dev))
        sde.set_model(net)
        out = sde.reverse_posterior(x0.to(dev), T=3, noise_tape=tape.to(dev), text_context=tc.to(dev), image_context=ic.to(dev))
        outs.append((eps.cpu(), out.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_bf16_activation_stream_variants(ops):
    """The IR-SDE UNet's bf16 activation stream: bf16 rows into ln_rows / add_rows / tail_conv3x3 / linattn_block, a bf16
    residual in the conv epilogue.  Each is checked against fp64 on the SAME bf16-rounded inputs (so only the output
    rounding and the bf16 MFMA operands remain)."""
    g = torch.Generator().manual_seed(77)
    bf = torch.bfloat16
    # ln_rows: bf16 x -> fp32 / bf16 y
    x = (torch.randn(37, 64 + 8, generator=g) * 2 + 1).to(bf)
    gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
    ref = F.layer_norm(x[:, :64].double(), (64,), gamma.double(), beta.double(), 1e-5)
    for odt in (torch.float32, bf):
        y = torch.full((37, 72), 3.0, device="cuda", dtype=odt)
        ops.ln_rows(x.cuda()[:, :64], gamma.cuda(), beta.cuda(), y[:, :64], 1e-5)
        assert err(y[:, :64], ref) < (1e-5 if odt == torch.float32 else 4e-2) and float(y[:, 64:].float().min()) == 3.0
    # add_rows: fp32 + bf16 -> bf16 / fp32
    a, b = torch.randn(9, 24, generator=g), torch.randn(9, 24, generator=g).to(bf)
    for odt in (torch.float32, bf):
        o = torch.zeros(9, 28, device="cuda", dtype=odt)
        ops.add_rows(a.cuda()[:, :20], b.cuda()[:, 4:], o[:, 8:])
        exp = (a[:, :20] + b[:, 4:].float()).to(odt)
        assert torch.equal(o[:, 8:].cpu(), exp) and float(o[:, :8].float().abs().max()) == 0.0
    # conv epilogue: bf16 residual (3x3 halo path and a plain 1x1 GEMM)
    bb, c, h, w, n = 2, 32, 20, 24, 64
    xi = torch.randn(bb, c, h, w, generator=g)
    wt = torch.randn(n, c, 3, 3, generator=g) / math.sqrt(9 * c)
    res = torch.randn(bb * h * w, n, generator=g).to(bf)
    xin = to_rows(xi).to(bf)
    xe = xin.float().cpu().reshape(bb, h, w, c).permute(0, 3, 1, 2).double()
    for odt in (torch.float32, bf):
        out = torch.empty(bb * h * w, n, device="cuda", dtype=odt)
        ops.conv_gemm(xin, wt.permute(0, 2, 3, 1).reshape(n, -1).bfloat16().cuda().contiguous(), out, B=bb, H=h, W=w, C_=c, N=n,
                      KH=3, KW=3, pad=1, act=ops.ACT_SILU, res=res.cuda())
        refc = F.silu(F.conv2d(xe, wt.bfloat16().double(), padding=1)) + from_rows(res.double(), bb, h, w)
        assert rel(from_rows(out, bb, h, w), refc) < (1e-5 if odt == torch.float32 else 1e-2)
    w1 = (torch.randn(128, 64, generator=g) / 8).to(bf)
    x1 = torch.randn(300, 64, generator=g).to(bf)
    r1 = torch.randn(300, 128, generator=g).to(bf)
    o1 = torch.empty(300, 128, device="cuda")
    ops.conv_gemm(x1.cuda(), w1.cuda(), o1, B=1, H=1, W=300, C_=64, N=128, res=r1.cuda())
    assert err(o1, x1.double() @ w1.double().T + r1.double()) < 1e-4
    with pytest.raises(Exception):   # bf16 residual rows must start on 8-byte boundaries
        ops.conv_gemm(x1.cuda(), w1.cuda(), o1, B=1, H=1, W=300, C_=64, N=128, res=r1.cuda()[:, 1:])   # misaligned view
    # tail conv on bf16 rows
    feat = torch.randn(2, 32, 16, 32, generator=g).to(bf)
    wt2, bt = torch.randn(3, 32, 3, 3, generator=g) * 0.1, torch.randn(3, generator=g)
    o = torch.empty(2, 3, 13, 18, device="cuda")
    ops.tail_conv3x3(to_rows(feat.float()).to(bf), 2, 16, 32, 32, wt2.permute(0, 2, 3, 1).contiguous().cuda(), bt.cuda(), 3, o)
    assert err(o, F.conv2d(feat.double(), wt2.double(), bt.double(), padding=1)[..., :13, :18]) < 5e-5   # MFMA, weights hi + lo
    for cin, hh, ww in ((64, 40, 50), (48, 16, 32)):   # 64: MFMA form with two chunks, ragged tiles; 48: the per-pixel form
        feat = torch.randn(1, cin, hh, ww, generator=g).to(bf)
        wt3, bt3 = torch.randn(4, cin, 3, 3, generator=g) * 0.1, torch.randn(4, generator=g)
        o = torch.empty(1, 4, hh - 3, ww - 1, device="cuda")
        ops.tail_conv3x3(to_rows(feat.float()).to(bf), 1, hh, ww, cin, wt3.permute(0, 2, 3, 1).contiguous().cuda(), bt3.cuda(), 4, o)
        assert err(o, F.conv2d(feat.double(), wt3.double(), bt3.double(), padding=1)[..., :hh - 3, :ww - 1]) < 5e-5, cin
    # fused LinearAttention block on bf16 rows: same function as on the fp32 rows holding the same values
    bq, nq, dim = 2, 300, 64
    xq = (torch.randn(bq * nq, dim, generator=g) * 1.5).to(bf)
    g_pre, g_out = torch.rand(dim, generator=g) + 0.5, torch.rand(dim, generator=g) + 0.5
    w_qkv = (torch.randn(384, dim, generator=g) * (2.0 / dim ** 0.5)).to(bf)
    w_out = (torch.randn(dim, 128, generator=g) * (1.0 / 128 ** 0.5)).to(bf)
    b_out = torch.randn(dim, generator=g) * 0.1
    ws = torch.empty(ops.linattn_block_workspace_bytes(bq, nq) // 4, device="cuda")
    o32 = torch.empty(bq * nq, dim, device="cuda")
    o16 = torch.full((bq * nq, dim + 8), 5.0, device="cuda", dtype=bf)
    args = (bq, nq, dim, g_pre.cuda(), w_qkv.cuda(), w_out.cuda(), b_out.cuda(), g_out.cuda())
    ops.linattn_block(xq.float().cuda(), *args, o32, ws)
    ops.linattn_block(xq.cuda(), *args, o16[:, :dim], ws)
    assert torch.equal(o16[:, :dim], o32.to(bf)) and float(o16[:, dim:].float().min()) == 5.0
    with pytest.raises(TypeError):
        ops.linattn_block(xq.cuda(), *args, o32, ws)


def test_unet_stem_7x7_mfma(ops):
    """The IR-SDE UNet's init conv (7x7, 8 zero-padded channels -> 64, fp32 rows in, bf16 rows out) has its own MFMA kernel:
    ragged tiles (16-pixel strips, 32-row blocks), output into a column slice of a wider buffer, against fp64 on the
    bf16-rounded operands; the generic implicit GEMM (fp32 output: not eligible) must agree."""
    g = torch.Generator().manual_seed(91)
    b, h, w, n = 2, 40, 50, 64
    x = torch.randn(b, 8, h, w, generator=g)
    x[:, 6:] = 0
    wt = torch.randn(n, 8, 7, 7, generator=g) / math.sqrt(49 * 6)
    wd = wt.permute(0, 2, 3, 1).reshape(n, -1).bfloat16().cuda().contiguous()
    out = torch.full((b * h * w, 2 * n), 3.0, device="cuda", dtype=torch.bfloat16)
    ops.conv_gemm(to_rows(x), wd, out[:, n:], B=b, H=h, W=w, C_=8, N=n, KH=7, KW=7, pad=3)
    ref = F.conv2d(x.bfloat16().double(), wt.bfloat16().double(), padding=3)
    assert rel(from_rows(out[:, n:].float(), b, h, w), ref) < 6e-3 and float(out[:, :n].float().min()) == 3.0
    o32 = torch.empty(b * h * w, n, device="cuda")
    ops.conv_gemm(to_rows(x), wd, o32, B=b, H=h, W=w, C_=8, N=n, KH=7, KW=7, pad=3)
    assert rel(from_rows(o32, b, h, w), ref) < 1e-4
    assert float((out[:, n:].float() - o32).abs().max()) < 0.05


@pytest.mark.parametrize("shape", [(3, 256, 192, 576), (2, 128, 64, 128), (16, 1024, 768, 1536),
                                   (16, 1024, 192, 576), (8, 512, 768, 2304), (5, 1280, 384, 1152)])   # the 256 x 288 kernel (round 3)
def test_fold_affine_weights_and_batched_gemm(ops, shape):
    """LayerNorm2d's per-image affine folded into the next 1x1 conv (S/utils.py:18-36 -> S/wtb.py:50,71):
    W (a_b x + s_b) + bias == (W diag(a_b)) x + (W s_b + bias), with per-image bf16 weight panels in the LDS-DMA GEMM."""
    b, n, k, nn = shape
    g = torch.Generator().manual_seed(41)
    x = torch.randn(b * n, k, generator=g).bfloat16()
    W = torch.randn(nn, k, generator=g) / math.sqrt(k)
    bias = torch.randn(nn, generator=g)
    a, s = torch.rand(b, k, generator=g) + 0.5, torch.randn(b, k, generator=g)
    Wb = torch.empty(b, nn, k, device="cuda", dtype=torch.bfloat16)
    bb = torch.empty(b, nn, device="cuda")
    ops.fold_affine_weights(W.cuda(), a.cuda(), s.cuda(), bias.cuda(), Wb, bb)
    assert err(Wb, W.double()[None] * a.double()[:, None, :]) < 2e-2            # bf16 rounding of O(1) entries
    assert err(bb, s.double() @ W.double().T + bias.double()) < 1e-4
    out = torch.empty(b * n, nn, device="cuda", dtype=torch.bfloat16)
    ops.conv_gemm(x.cuda(), Wb, out, B=b, H=n // 16, W=16, C_=k, N=nn, batch_bias=bb, act=ops.ACT_GELU, w_batched=True)
    xn = x.double().reshape(b, n, k) * a.double()[:, None] + s.double()[:, None]
    ref = F.gelu(xn @ W.double().T + bias.double())
    assert rel(out.reshape(b, n, nn), ref) < 1e-2
    # exact restatement on the bf16 panels: only the accumulation order and the output rounding remain
    ref2 = F.gelu(torch.einsum("bnk,bok->bno", x.double().reshape(b, n, k), Wb.double().cpu()) + bb.double().cpu()[:, None])
    assert rel(out.reshape(b, n, nn), ref2) < 5e-3
    if nn % 288 == 0 and b * n >= 4096:   # taken by the 256 x 288 kernel: against the 128 x 128 LDS-DMA kernel (variant bit 6)
        lib = ops._lib.load()
        out2 = torch.empty_like(out)
        lib.uwir_conv_variant(64)
        try:
            ops.conv_gemm(x.cuda(), Wb, out2, B=b, H=n // 16, W=16, C_=k, N=nn, batch_bias=bb, act=ops.ACT_GELU, w_batched=True)
        finally:
            lib.uwir_conv_variant(0)
        assert rel(out2.reshape(b, n, nn), ref2) < 5e-3 and float((out.float() - out2.float()).abs().max()) < 0.07
    with pytest.raises(Exception):      # tiles would straddle images
        ops.conv_gemm(x.cuda()[:b * 96], Wb, out[:b * 96], B=b, H=6, W=16, C_=k, N=nn, batch_bias=bb, w_batched=True)


def test_precise_tail_path_variants(ops):
    """The kernels behind the UNet's precise tail path (bf16 mode, unet.precise_tail): the 7x7 stem MFMA kernel writing
    fp32 rows, the resident-weight 3x3 conv (C = 64, bf16 rows) with an fp32 residual and fp32 / bf16 output, and the tail
    conv on fp32 rows (inputs and weights split hi + lo in the kernel: products to ~2^-16)."""
    g = torch.Generator().manual_seed(93)
    bf = torch.bfloat16
    # stem -> fp32 rows in a column slice (the stem half of the UNet's catF)
    b, h, w, n = 2, 40, 50, 64
    x = torch.randn(b, 8, h, w, generator=g)
    x[:, 6:] = 0
    wt = torch.randn(n, 8, 7, 7, generator=g) / math.sqrt(49 * 6)
    wd = wt.permute(0, 2, 3, 1).reshape(n, -1).bfloat16().cuda().contiguous()
    out = torch.full((b * h * w, 2 * n), 3.0, device="cuda")
    ops.conv_gemm(to_rows(x), wd, out[:, n:], B=b, H=h, W=w, C_=8, N=n, KH=7, KW=7, pad=3)
    ref = F.conv2d(x.bfloat16().double(), wt.bfloat16().double(), padding=3)
    assert rel(from_rows(out[:, n:], b, h, w), ref) < 1e-5 and float(out[:, :n].min()) == 3.0
    o16 = torch.empty(b * h * w, n, device="cuda", dtype=bf)
    ops.conv_gemm(to_rows(x), wd, o16, B=b, H=h, W=w, C_=8, N=n, KH=7, KW=7, pad=3)
    assert torch.equal(o16, out[:, n:].to(bf))                  # same accumulators, rounded at the store
    # resident-weight conv (C = 64 bf16 rows): fp32 residual, per-sample scale / bias, SiLU, fp32 and bf16 rows out
    bb, c, hh, ww = 3, 64, 40, 52
    xi = to_rows(torch.randn(bb, c, hh, ww, generator=g)).to(bf)
    wc = torch.randn(64, c, 3, 3, generator=g) / math.sqrt(9 * c)
    res32 = torch.randn(bb * hh * ww, 64 + 8, generator=g).cuda()
    sc, sh = (torch.rand(bb, 64, generator=g) + 0.5).cuda(), torch.randn(bb, 64, generator=g).cuda()
    xe = xi.float().cpu().reshape(bb, hh, ww, c).permute(0, 3, 1, 2).double()
    refc = F.conv2d(xe, wc.bfloat16().double(), padding=1) * sc.cpu().double()[:, :, None, None] + sh.cpu().double()[:, :, None, None]
    refc = F.silu(refc) + from_rows(res32[:, :64].double(), bb, hh, ww).cpu()
    wcd = wc.permute(0, 2, 3, 1).reshape(64, -1).bfloat16().cuda().contiguous()
    outs = {}
    for odt in (torch.float32, bf):
        o = torch.empty(bb * hh * ww, 64, device="cuda", dtype=odt)
        ops.conv_gemm(xi, wcd, o, B=bb, H=hh, W=ww, C_=c, N=64, KH=3, KW=3, pad=1, act=ops.ACT_SILU, res=res32[:, :64],
                      batch_scale=sc, batch_bias=sh)
        assert rel(from_rows(o, bb, hh, ww), refc) < (1e-5 if odt == torch.float32 else 1e-2), odt
        outs[odt] = o
    from underwater_image_restoration_amd import _lib
    _lib.load().uwir_conv_variant(1 << 3)                      # LDS-DMA convs off: the register-staged kernel must agree
    try:
        o = torch.empty(bb * hh * ww, 64, device="cuda")
        ops.conv_gemm(xi, wcd, o, B=bb, H=hh, W=ww, C_=c, N=64, KH=3, KW=3, pad=1, act=ops.ACT_SILU, res=res32[:, :64],
                      batch_scale=sc, batch_bias=sh)
    finally:
        _lib.load().uwir_conv_variant(0)
    assert float((o - outs[torch.float32]).abs().max()) < 1e-4
    # tail conv on fp32 rows: MFMA with hi + lo splits of both operands, against fp64 on the UNROUNDED operands
    for cin, h2, w2, co in ((64, 40, 50, 3), (32, 16, 32, 4)):
        feat = torch.randn(2, cin, h2, w2, generator=g) * 2
        wt3, bt3 = torch.randn(co, cin, 3, 3, generator=g) * 0.1, torch.randn(co, generator=g)
        rows = torch.full((2 * h2 * w2, cin + 8), 9.0, device="cuda")
        rows[:, :cin] = to_rows(feat)
        o = torch.empty(2, co, h2 - 3, w2 - 1, device="cuda")
        ops.tail_conv3x3(rows[:, :cin], 2, h2, w2, cin, wt3.permute(0, 2, 3, 1).contiguous().cuda(), bt3.cuda(), co, o, split=True)
        refd = F.conv2d(feat.double(), wt3.double(), bt3.double(), padding=1)[..., :h2 - 3, :w2 - 1]
        e = rel(o, refd)
        assert e < 5e-5, (cin, e)                                # relative to max |out|; plain bf16 operands give ~4e-3


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_conv_gemm_geglu_store(ops, prec):
    """GEGLU (attention.py:37-44) in the projection's epilogue: weight rows in groups of 8 values + their 8 gates, output
    [M, F] = value * gelu(gate); must equal the projection followed by the stand-alone geglu kernel."""
    g = torch.Generator().manual_seed(93)
    M, K, F_ = 300, 64, 96
    x = torch.randn(M, K, generator=g)
    w = torch.randn(2 * F_, K, generator=g) / 8
    b = torch.randn(2 * F_, generator=g)
    pair = lambda t: torch.stack([t[:F_].reshape(F_ // 8, 8, *t.shape[1:]), t[F_:].reshape(F_ // 8, 8, *t.shape[1:])], 1).reshape(t.shape)
    dt = _wdt(prec)
    xd, wd = x.to(dt).cuda(), pair(w).to(dt).cuda().contiguous()
    out = torch.full((M, F_ + 8), 2.0, device="cuda", dtype=dt)
    ops.conv_gemm(xd, wd, out[:, :F_], B=1, H=1, W=M, C_=K, N=2 * F_, bias=pair(b).cuda().contiguous(), store_mode=2)
    h = xd.double().cpu() @ w.to(dt).double().T + b.double()
    ref = h[:, :F_] * F.gelu(h[:, F_:])
    e, r = err(out[:, :F_], ref), rel(out[:, :F_].float().cpu(), ref)
    print(f"\n[geglu store {prec}] max abs {e:.3e} rel {r:.3e} (|ref|max {float(ref.abs().max()):.2f})")
    assert (e < 2e-5 if prec == "fp32" else r < 5e-3) and float(out[:, F_:].float().min()) == 2.0
    with pytest.raises(Exception):
        ops.conv_gemm(xd, wd, out[:, :F_], B=1, H=1, W=M, C_=K, N=2 * F_, store_mode=2, act=ops.ACT_GELU)


def test_bf16_subband_stream_variants(ops):
    """The student's bf16 sub-band stream: dwt2 writing bf16 sub-bands, idwt2 reading bf16 hf, LayerNorm2d statistics and
    the norm-and-cast pass on bf16 rows, the lean residual GEMM class starting from bf16 rows (plain and affine).  Each
    against fp64 on the same (bf16-rounded where stored as bf16) values."""
    g = torch.Generator().manual_seed(95)
    bf = torch.bfloat16
    b, c, h, w = 2, 8, 12, 20
    x = torch.randn(b, c, h, w, generator=g)
    ll, lh, hl, hh = R.haar_dwt2(x.double())
    h2, w2 = h // 2, w // 2
    dll = torch.full((b * h2 * w2, c + 8), 5.0, device="cuda", dtype=bf)
    dhf = torch.empty(b * h2 * w2, 3 * c, device="cuda", dtype=bf)
    ops.dwt2(to_rows(x), b, h, w, c, dll[:, :c], dhf)
    assert torch.equal(from_rows(dll[:, :c].float(), b, h2, w2).cpu(), ll.float().to(bf).float()) or \
        err(from_rows(dll[:, :c].float(), b, h2, w2), ll) < 2e-2
    assert err(from_rows(dhf.float(), b, h2, w2), torch.cat([lh, hl, hh], 1)) < 2e-2 and float(dll[:, c:].float().min()) == 5.0
    with pytest.raises(TypeError):
        ops.dwt2(to_rows(x), b, h, w, c, dll[:, :c], dhf.float())
    out = torch.empty(b * 4 * h2 * w2, c, device="cuda")
    llf = to_rows(ll.float())
    ops.idwt2(llf, dhf, b, h2, w2, c, out)
    hfd = from_rows(dhf.float(), b, h2, w2).double().cpu()
    assert err(from_rows(out, b, 2 * h2, 2 * w2), R.haar_idwt2(ll, hfd[:, :c], hfd[:, c:2 * c], hfd[:, 2 * c:])) < 1e-5
    # LayerNorm2d statistics + folded affine on bf16 rows == on the fp32 rows holding the same values
    B, n, C = 3, 700, 48
    xr = (torch.randn(B * n, C, generator=g) * 2 + 0.5).to(bf)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    res = {}
    for dt in (torch.float32, bf):
        a, s, mean = (torch.empty(B, C, device="cuda") for _ in range(3))
        ws = torch.empty(max(ops.ln2d_workspace_bytes(B, n, C) // 8, 1), dtype=torch.float64, device="cuda")
        ops.ln2d_affine(xr.to(dt).cuda(), B, n, C, 1e-6, gamma.cuda(), beta.cuda(), None, a, s, mean, None, ws)
        res[dt] = (a.clone(), s.clone(), mean.clone())
    for u, v in zip(res[torch.float32], res[bf]):   # (bf16 rows take 16-byte loads: another fixed summation order)
        assert torch.allclose(u, v, rtol=3e-6, atol=3e-6)
    xd = xr.double().reshape(B, n, C)
    assert err(res[bf][2], xd.mean(1)) < 1e-5
    rstd = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-6)
    assert err(res[bf][0], rstd * gamma.double()) < 1e-5 and err(res[torch.float32][0], rstd * gamma.double()) < 1e-5
    # norm-and-cast from bf16 rows
    a, s = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
    o = torch.empty(B * n, C, device="cuda", dtype=bf)
    ops.affine_cast(xr.cuda(), B, n, C, a.cuda(), s.cuda(), o)
    assert err(o.reshape(B, n, C), xd * a.double()[:, None] + s.double()[:, None]) < 5e-2
    # lean residual class (LDS-DMA GEMM, no activation): bf16 residual rows, plain and affine
    M, K, N = B * 256, 192, 192
    A = torch.randn(M, K, generator=g).to(bf)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf)
    bias, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g).to(bf)
    rs, rh = torch.rand(B, N, generator=g) + 0.5, torch.randn(B, N, generator=g)
    base = A.double() @ Wt.double().T + bias.double()
    for affine in (False, True):
        for odt in (torch.float32, bf):
            out = torch.empty(M, N, device="cuda", dtype=odt)
            kw = dict(res_scale=rs.cuda(), res_shift=rh.cuda()) if affine else {}
            ops.conv_gemm(A.cuda(), Wt.cuda(), out, B=B, H=16, W=16, C_=K, N=N, bias=bias.cuda(), res=r.cuda(), **kw)
            rr = r.double().reshape(B, 256, N)
            ref = base + ((rr * rs.double()[:, None] + rh.double()[:, None]) if affine else rr).reshape(M, N)
            assert err(out, ref) < (1e-4 if odt == torch.float32 else 6e-2), (affine, odt)
