"""Stem / head kernels of the student at B = 16, 256 x 256, bf16 rows: MFMA kernels vs the per-pixel ones (A/B/A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib, ops
lib = _lib.load()
B, H, W = 16, 256, 256
g = torch.Generator().manual_seed(0)
x = torch.rand(B, 3, H, W, generator=g).cuda()
ws, bs = (torch.randn(3, 3, 3, 64, generator=g) * 0.2).cuda(), torch.randn(64, generator=g).cuda()
rows = torch.empty(B * H * W, 64, device="cuda", dtype=torch.bfloat16)
wh, bh = (torch.randn(3, 3, 3, 64, generator=g) * 0.05).cuda(), torch.randn(3, generator=g).cuda()
a, s = (torch.rand(B, 64, generator=g) + 0.5).cuda(), torch.randn(B, 64, generator=g).cuda()
out = torch.empty(B, 3, H, W, device="cuda")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(2):
    for var in (1, 2):
        lib.uwir_head_variant(var)
        ts = t(lambda: ops.stem_conv3x3(x, ws, bs, rows))
        th = t(lambda: ops.head_conv3x3(rows, B, H, W, 64, a, s, wh, bh, 3, x, out, None))
        print(f"variant {var}: stem {ts:.1f} us ({rows.numel() * 2 / ts / 1e6:.2f} TB/s of output), head {th:.1f} us "
              f"({rows.numel() * 2 / th / 1e6:.2f} TB/s of input)")
lib.uwir_head_variant(1)
