import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
for _ in range(100): a @ b
bf = torch.bfloat16
for B, n, dim in ((8, 65536, 64), (8, 16384, 64), (8, 4096, 128), (8, 1024, 256)):
    x = torch.randn(B * n, dim, device="cuda").to(bf)
    out = torch.empty_like(x)
    g1, g2, bo = torch.ones(dim, device="cuda"), torch.ones(dim, device="cuda"), torch.zeros(dim, device="cuda")
    wq = (torch.randn(384, dim, device="cuda") / dim ** 0.5).to(bf)
    wo = (torch.randn(dim, 128, device="cuda") / 128 ** 0.5).to(bf)
    ws = torch.empty(ops.linattn_block_workspace_bytes(B, n) // 4, device="cuda")
    f = lambda: ops.linattn_block(x, B, n, dim, g1, wq, wo, bo, g2, out, ws)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    print(f"B{B} n{n} dim{dim}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us")
