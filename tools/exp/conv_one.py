import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from underwater_image_restoration_amd import _lib, ops
lib = _lib.load(); lib.uwir_conv_variant(0)
B, S, C, N = 8, 256, 64, 64
g = torch.Generator().manual_seed(0)
M = B * S * S
x = torch.randn(M, C, generator=g).cuda()
w = (torch.randn(N, 9 * C, generator=g) / 24).cuda().bfloat16()
ss = torch.randn(B, 2 * N, generator=g).cuda()
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.conv_gemm(x, w, out, B=B, H=S, W=S, C_=C, N=N, KH=3, KW=3, pad=1, act=ops.ACT_SILU, batch_scale=ss[:, :N], batch_bias=ss[:, N:])
torch.cuda.synchronize()
