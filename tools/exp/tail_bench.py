import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
B, H, C = 8, 256, 64
x = torch.randn(B * H * H, C, device="cuda").bfloat16()
w = (torch.randn(3, 3, 3, C, device="cuda") * 0.1).contiguous()
b = torch.randn(3, device="cuda")
o = torch.empty(B, 3, H, H, device="cuda")
for dt in (torch.bfloat16, torch.float32):
    xx = x.to(dt)
    f = lambda: ops.tail_conv3x3(xx, B, H, H, C, w, b, 3, o)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(dt, f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
