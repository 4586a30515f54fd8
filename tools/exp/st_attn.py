"""SpatialTransformer self-attention of the IR-SDE UNet (n = 1024, hd = 32): LDS kernel vs packed stream (uwir_attn_variant(8))."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib, ops  # noqa: E402

lib = _lib.load()
v = int(sys.argv[1]) if len(sys.argv) > 1 else -1
lib.uwir_attn_variant(v)
a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
for _ in range(200):
    a @ b
for B, n, heads in ((8, 1024, 16), (8, 1024, 8), (8, 4096, 8)):
    dim = heads * 32
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B * n, 3 * dim, generator=g) * 0.5).cuda().bfloat16()
    qkv[:, :dim] *= 1.4426950408889634 / 32 ** 0.5
    out = torch.empty(B * n, dim, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.flash_attn(qkv[:, :dim], qkv[:, dim:2 * dim], qkv[:, 2 * dim:], None, B, n, heads, 32, 0.0, out, None)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    q = qkv[:, :dim].float().reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    k = qkv[:, dim:2 * dim].float().reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    vv = qkv[:, 2 * dim:].float().reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * 0.6931471805599453, -1) @ vv).permute(0, 2, 1, 3).reshape(B * n, dim)
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"variant {v}: B{B} n{n} heads{heads}: {us:.1f} us  {4.0 * B * heads * n * n * 32 / us / 1e6:.0f} TF/s  "
          f"max err {float((out.float() - ref).abs().max()):.2e}", flush=True)
