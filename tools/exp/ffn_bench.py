"""WTB feed-forward block: one launch (ops.ffn_fused) against the two GEMM launches, at the student's levels 0 and 1 (B = 16, 256^2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
bf = torch.bfloat16

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

a = torch.randn(4096, 4096, device="cuda")
for _ in range(100): a @ a
for C, n in ((192, 16384), (384, 4096)):
    B = 16; M = B * n
    x = torch.randn(M, C, device="cuda").to(bf)
    w0b = (torch.randn(B, 2 * C, C, device="cuda") / C ** 0.5).to(bf)
    bias0 = torch.randn(B, 2 * C, device="cuda") * 0.3
    w2 = (torch.randn(C, 2 * C, device="cuda") / (2 * C) ** 0.5).to(bf); b2 = torch.randn(C, device="cuda")
    w2p = ops.ffn_permute_w2(w2)
    hidden = torch.empty(M, 2 * C, device="cuda", dtype=bf); out = torch.empty(M, C, device="cuda", dtype=bf)
    def two():
        ops.conv_gemm(x, w0b, hidden, B=B, H=1, W=n, C_=C, N=2 * C, batch_bias=bias0, act=ops.ACT_GELU, w_batched=True)
        ops.conv_gemm(hidden, w2, out, B=B, H=1, W=n, C_=2 * C, N=C, bias=b2, res=x)
    def one():
        ops.ffn_fused(x, w0b, bias0, w2p, b2, B, n, C, out)
    r = [(timeit(two), timeit(one)) for _ in range(3)]
    fl = 2.0 * M * C * 2 * C * 2
    print(f"C{C} n{n}: two launches " + "/".join(f"{t[0]:.1f}" for t in r) + " us, one launch " + "/".join(f"{t[1]:.1f}" for t in r)
          + f" us ({fl / min(t[1] for t in r) / 1e6:.0f} TF/s)", flush=True)
