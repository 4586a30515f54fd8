"""Ablations of the 256 x 288 large-K GEMM through uwir_gemm_wide_probe (include/uwir.h): 0 = the product kernel, 1 = every
wave issues its fill early, 2 = mid-stage, 3 = no MFMAs, 4 = no fill inside the loop, 5 = the fill alone; and the product
kernel against the 256 x 192 one (uwir_conv_variant bit 6) through uwir_conv_gemm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops, _lib
lib = _lib.load()

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
st = torch.cuda.current_stream().cuda_stream
for M, N, K in ((262144, 576, 1728), (65536, 1152, 3456), (16384, 2304, 6912)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def conv(v):
        def f():
            lib.uwir_conv_variant(v)
            ops.conv_gemm(A, W, o, B=1, H=1, W=M, C_=K, N=N, bias=bias)
        return f
    def probe(v):
        def f():
            assert lib.uwir_gemm_wide_probe(A.data_ptr(), W.data_ptr(), o.data_ptr(), bias.data_ptr(), M, N, K, v, st) == 0
        return f
    pv = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(6))
    arms = [("conv_gemm 256x192", conv(64)), ("conv_gemm 256x288", conv(0))] + [(f"probe {v}", probe(v)) for v in pv]
    res = {k: [] for k, _ in arms}
    for _ in range(3):
        for k, f in arms: res[k].append(timeit(f))
    lib.uwir_conv_variant(0)
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}")
    for k, _ in arms:
        print(f"    {k:20s} " + " ".join(f"{t:.3f}" for t in res[k]) + f" ms  ({fl / min(res[k]) / 1e9:.0f} TF/s-equivalent)", flush=True)
