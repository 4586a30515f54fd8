import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
for _ in range(100): a @ b
for B, n, C in ((16, 16384, 64), (16, 16384, 192), (16, 4096, 128), (16, 4096, 384), (16, 1024, 256), (16, 1024, 768), (16, 65536, 64)):
    dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
    x = torch.randn(B * n, C, device="cuda").to(dt)
    g, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ao, so = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
    ws = torch.empty(max(ops.ln2d_workspace_bytes(B, n, C) // 8, 1), dtype=torch.float64, device="cuda")
    f = lambda: ops.ln2d_affine(x, B, n, C, 1e-6, g, bt, None, ao, so, None, None, ws)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B{B} n{n} C{C}: {us:.1f} us  {x.numel() * x.element_size() / us / 1e6:.2f} TB/s (statistics + finalize)")
