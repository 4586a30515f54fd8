"""SpatialTransformer-sized GEMMs (M = 8192 rows): the 256-tile launch vs the few-tile paths (M = 8064 -> 252 tiles takes the
double-buffered / split-K kernel): is the dispatch threshold right?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
for (N, K, odt, res) in ((512, 512, torch.float32, True), (1536, 512, torch.bfloat16, False), (512, 2048, torch.bfloat16, True), (4096, 512, torch.bfloat16, False), (256, 256, torch.float32, True), (768, 256, torch.bfloat16, False)):
    line = f"N{N} K{K} out {str(odt)[6:]} res={res}:"
    for M in (8192, 8064, 8192, 8064):
        A = torch.randn(M, K, generator=g).cuda().bfloat16(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=odt); bias = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).cuda() if res else None
        t = timeit(lambda: ops.conv_gemm(A, W, out, B=8, H=M // 8 // 32, W=32, C_=K, N=N, bias=bias, res=r))
        line += f"  M{M}: {t:.1f} us"
    print(line, flush=True)
