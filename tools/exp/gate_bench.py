"""WFIGate + idwt2: one launch (ops.gate_idwt) against the three launches, at the student's three levels (B = 16, 256^2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops
bf = torch.bfloat16

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

a = torch.randn(4096, 4096, device="cuda"); 
for _ in range(100): a @ a
for C, hw in ((64, 128), (128, 64), (256, 32)):
    B, h, w = 16, hw, hw
    M, Kg = B * h * w, 2 * C + 16
    gi = torch.randn(M, Kg, device="cuda")
    w1 = (torch.randn(64, Kg, device="cuda") / Kg ** 0.5).to(bf); w2 = (torch.randn(C, 64, device="cuda") / 8).to(bf)
    bb, b2 = torch.randn(B, 64, device="cuda"), torch.randn(C, device="cuda")
    hf = torch.randn(M, 3 * C, device="cuda").to(bf)
    ghid = torch.empty(M, 64, device="cuda", dtype=bf)
    alpha, merged = torch.empty(B, h, w, C, device="cuda"), torch.empty(M, C, device="cuda")
    out = torch.empty(4 * M, C, device="cuda", dtype=bf)
    def three():
        ops.conv_gemm(gi, w1, ghid, B=B, H=h, W=w, C_=Kg, N=64, batch_bias=bb, act=ops.ACT_RELU)
        ops.conv_gemm(ghid, w2, alpha.view(M, C), B=B, H=h, W=w, C_=64, N=C, bias=b2, act=ops.ACT_SIGMOID,
                      blend_t=gi[:, :C], blend_f=gi[:, C:2 * C], blend_out=merged)
        ops.idwt2(merged, hf, B, h, w, C, out)
    def one():
        ops.gate_idwt(gi, w1, w2, bb, b2, hf, B, h, w, C, out, alpha, None)
    r = [(timeit(three), timeit(one)) for _ in range(3)]
    by = M * (Kg * 4 + 3 * C * 2 + C * 4 + 4 * C * 2)
    print(f"C{C} {hw}x{hw}: three launches " + "/".join(f"{x[0]:.1f}" for x in r) + " us, one launch " + "/".join(f"{x[1]:.1f}" for x in r)
          + f" us ({by / min(x[1] for x in r) / 1e3:.0f} GB/s of its {by / 1e6:.0f} MB)", flush=True)
