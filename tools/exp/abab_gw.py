"""gemm_wide variants A/B/A/B (uwir_conv_variant bits 6-7) + agreement of the results."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops, _lib
lib = _lib.load()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 64]

def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

a = torch.randn(8192, 8192, device="cuda").bfloat16()
for _ in range(50): a @ a
torch.cuda.synchronize()
for M, N, K in ((262144, 576, 1728), (65536, 1152, 3456), (16384, 2304, 6912)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); outs = {}
    f = None
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            lib.uwir_conv_variant(v)
            res[v].append(timeit(lambda: ops.conv_gemm(A, W, out, B=1, H=1, W=M, C_=K, N=N, bias=bias)))
            outs[v] = out
    lib.uwir_conv_variant(0)
    fl = 2.0 * M * N * K
    line = f"GEMM M{M} N{N} K{K}: " + " | ".join(f"v{v}: " + " ".join(f"{t:.3f}" for t in res[v]) + f" ms ({fl / min(res[v]) / 1e9:.0f} TF/s)" for v in variants)
    eq = all(torch.equal(outs[variants[0]], outs[v]) for v in variants[1:])
    ref = (A[:512].float() @ W.float().T + bias)
    err = float((outs[variants[-1]][:512].float() - ref).abs().max())
    print(line, "| results bit-equal:", eq, f"| max err vs fp32 ref (512 rows) {err:.3e}")
