"""Micro-benchmark of uwir_dwconv_ms at the student's three level shapes (B 16, bf16 in / out):
    python tools/bench_dwconv.py [reps] [tiles per block, comma separated; 0 = auto]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for B, H, W, C in ((16, 128, 128, 576), (16, 64, 64, 1152), (16, 32, 32, 2304)):
  for nt in ([0] if len(sys.argv) < 3 else [int(v) for v in sys.argv[2].split(",")]):
    from underwater_image_restoration_amd import _lib
    _lib.load().uwir_dwconv_tiles_per_block(nt)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * H * W, C, generator=g).cuda().bfloat16()
    taps = [torch.randn(k * k, C, generator=g).cuda() / k for k in (3, 5, 7)]
    biases = [torch.randn(C, generator=g).cuda() for _ in range(3)]
    out = torch.empty(B * H * W, 3 * C, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.dwconv_ms(x, B, H, W, C, taps, biases, out)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"nt={nt} dwconv_ms B{B} {H}x{W} C{C}: {ms:.3f} ms  {(x.numel() + out.numel()) * 2 / ms / 1e6:.0f} GB/s  "
          f"{2.0 * B * H * W * C * 83 / ms / 1e9:.1f} TF/s")
