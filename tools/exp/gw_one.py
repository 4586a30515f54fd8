"""A few launches of the merge GEMM of level 0 (M262144 N576 K1728) for PMC passes: python tools/exp/gw_one.py [reps] [variant]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops, _lib
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib.uwir_conv_variant(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
M, N, K = 262144, 576, 1728
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    ops.conv_gemm(A, W, out, B=1, H=1, W=M, C_=K, N=N, bias=bias)
torch.cuda.synchronize()
