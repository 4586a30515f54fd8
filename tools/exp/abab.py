"""A/B/A/B timing (order effects: the first configuration measured in a process can read 10-15 % slow)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops, _lib
lib = _lib.load()

def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

# warm the chip
a = torch.randn(8192, 8192, device="cuda").bfloat16()
for _ in range(50): a @ a
torch.cuda.synchronize()
for M, N, K in ((262144, 576, 1728), (65536, 1152, 3456), (16384, 2304, 6912)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.conv_gemm(A, W, out, B=1, H=1, W=M, C_=K, N=N, bias=bias)
    r = []
    for v in (2, 0, 2, 0, 2, 0):
        lib.uwir_conv_variant(v); r.append(timeit(f))
    lib.uwir_conv_variant(0)
    print(f"GEMM M{M} N{N} K{K}: 128x128 {r[0]:.3f} {r[2]:.3f} {r[4]:.3f} | 256x192 {r[1]:.3f} {r[3]:.3f} {r[5]:.3f} ms")
B, H, Wd, C = 16, 128, 128, 576
x = torch.randn(B * H * Wd, C, device="cuda").bfloat16()
taps = [torch.randn(k * k, C, device="cuda") / k for k in (3, 5, 7)]; biases = [torch.randn(C, device="cuda") for _ in range(3)]
o = torch.empty(B * H * Wd, 3 * C, device="cuda", dtype=torch.bfloat16)
f = lambda: ops.dwconv_ms(x, B, H, Wd, C, taps, biases, o)
r = []
for nt in (1, 4, 1, 4, 1, 4):
    lib.uwir_dwconv_tiles_per_block(nt); r.append(timeit(f))
lib.uwir_dwconv_tiles_per_block(0)
print(f"dwconv L0: nt=1 {r[0]:.3f} {r[2]:.3f} {r[4]:.3f} | nt=4 {r[1]:.3f} {r[3]:.3f} {r[5]:.3f} ms")
import math
b, n, heads, hd = 16, 16384, 4, 16
q, k, v1, v2 = ((torch.randn(b * n, heads * hd, device="cuda") * 0.5).bfloat16() for _ in range(4))
o1 = torch.empty(b * n, heads * hd, device="cuda", dtype=torch.bfloat16); o2 = torch.empty_like(o1)
f = lambda: ops.flash_attn(q, k, v1, v2, b, n, heads, hd, 1 / math.sqrt(hd), o1, o2)
r = []
for v in (0, 2, 0, 2, 0, 2, 6, 2, 6, 2):
    lib.uwir_attn_variant(v); r.append(timeit(f, 10))
lib.uwir_attn_variant(-1)
print("attn 2V hd16: ones-MFMA(QG4) " + " ".join(f"{t:.3f}" for t in r[0:6:2]) + " | selector(QG4) " + " ".join(f"{t:.3f}" for t in r[1:6:2])
      + " | selector QG2: " + " ".join(f"{t:.3f}" for t in r[6::2]) + " vs QG4 " + " ".join(f"{t:.3f}" for t in r[7::2]))
for B, H, Wd, C in ((16, 128, 128, 64), (16, 64, 64, 128)):
    x = torch.randn(B * H * Wd, C, device="cuda"); add = torch.randn(B * H * Wd, C, device="cuda")
    ws = torch.empty(ops.fft_workspace_bytes(B, H, Wd, C) // 4, device="cuda")
    amp, ph, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    f = lambda: (ops.fft2_amp_phase(x, B, H, Wd, C, amp, ph, ws), ops.ifft2_polar_add(amp, ph, add, B, H, Wd, C, out, ws))
    r = []
    for v in (4, 0, 4, 0, 4, 0):
        lib.uwir_conv_variant(v); r.append(timeit(f))
    lib.uwir_conv_variant(0)
    print(f"fft2 + ifft2 {H}x{Wd} C{C}: radix-2 {r[0]:.3f} {r[2]:.3f} {r[4]:.3f} | radix-4 {r[1]:.3f} {r[3]:.3f} {r[5]:.3f} ms")
