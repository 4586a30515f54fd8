"""How much of a foreign kernel hides under the level-0 attention launch?  Attention (B16, n16384, 12 heads, hd16) on one
stream, R launches of another kernel on a second stream; times alone and together, with and without the occupancy cap."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda")
B, n, heads, hd = 16, 16384, 12, 16
M = B * n
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(M, 3 * heads * hd, generator=g) * 0.5).to(dev, torch.bfloat16)
q, k, v = qkv[:, :192], qkv[:, 192:384], qkv[:, 384:]
o = torch.empty(M, 192, device=dev, dtype=torch.bfloat16)
A1728 = torch.randn(M, 1728, generator=g).to(dev, torch.bfloat16)
W1728 = (torch.randn(576, 1728, generator=g) * 0.02).to(dev, torch.bfloat16)
o576 = torch.empty(M, 576, device=dev, dtype=torch.bfloat16)
A192 = torch.randn(M, 192, generator=g).to(dev, torch.bfloat16)
W192 = (torch.randn(576, 192, generator=g) * 0.05).to(dev, torch.bfloat16)
w3, w5, w7 = (torch.randn(kk * kk, 576, generator=g).to(dev) * 0.1 for kk in (3, 5, 7))
b3 = torch.zeros(576, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def attn():
    ops.flash_attn(q, k, v, None, B, n, heads, hd, hd ** -0.5, o, None)


def gemm_big():
    ops.conv_gemm(A1728, W1728, o576, B=B, H=128, W=128, C_=1728, N=576)


def gemm_small():
    ops.conv_gemm(A192, W192, o576, B=B, H=128, W=128, C_=192, N=576)


def dw():
    ops.dwconv_ms(o576, B, 128, 128, 576, (w3, w5, w7), (b3, b3, b3), A1728)


def timed(fa, fb, ra, rb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream())
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        for _ in range(ra):
            fa()
    with torch.cuda.stream(s2):
        for _ in range(rb):
            fb()
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for name, fb, rb in (("gemm K1728", gemm_big, 8), ("gemm K192", gemm_small, 24), ("dwconv C576", dw, 8)):
    for cap in (8, 24):
        lib.uwir_attn_variant(cap)
        for _ in range(2):
            timed(attn, fb, 1, rb)
        ta = min(timed(attn, fb, 2, 0) for _ in range(3))
        tb = min(timed(attn, fb, 0, rb) for _ in range(3))
        tab = min(timed(attn, fb, 2, rb) for _ in range(3))
        print(f"{name:12s} cap={'on ' if cap == 24 else 'off'}: attn x2 {ta:.2f} ms, other x{rb} {tb:.2f} ms, together {tab:.2f} ms "
              f"(sum {ta + tb:.2f}; hidden {ta + tb - tab:.2f})", flush=True)
lib.uwir_attn_variant(-1)
