"""Production shapes of the IR-SDE UNet's 3x3 convs (B = 8, 256^2 ... 32^2), default kernel choice, 12 runs each: results
must be bit-identical between runs and within bf16 rounding of F.conv2d (checked on a sub-batch in fp64)."""
import math, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops
g = torch.Generator().manual_seed(9)
bad = 0
for (H, C, N, ups) in ((256, 64, 64, 0), (256, 128, 64, 0), (128, 128, 128, 0), (128, 192, 128, 0), (64, 256, 256, 0), (64, 384, 256, 0),
                       (32, 512, 512, 0), (32, 768, 512, 0), (128, 128, 64, 1), (64, 256, 128, 1), (32, 512, 256, 1), (128, 64, 64, 0)):
    B = 8
    x = torch.randn(B * H * H, C, generator=g).bfloat16().cuda()
    wt = (torch.randn(N, C, 3, 3, generator=g) / math.sqrt(9 * C)).bfloat16()
    wd = wt.permute(0, 2, 3, 1).reshape(N, -1).cuda().contiguous()
    bias = torch.randn(N, generator=g).cuda()
    OH = H << ups
    res = torch.randn(B * OH * OH, N, generator=g).bfloat16().cuda()
    outs = []
    for rep in range(12):
        out = torch.full((B * OH * OH, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.conv_gemm(x, wd, out, B=B, H=H, W=H, C_=C, N=N, KH=3, KW=3, pad=1, ups=ups, bias=bias, res=res)
        outs.append(out)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    xe = x[:H * H].double().reshape(1, H, H, C).permute(0, 3, 1, 2)
    if ups:
        xe = F.interpolate(xe, scale_factor=2, mode="nearest")
    ref = F.conv2d(xe, wt.double().cuda(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(OH * OH, N) + res[:OH * OH].double()
    e = float((outs[0][:OH * OH].double() - ref).norm() / ref.norm())
    ok = same and e < 1e-2
    bad += 0 if ok else 1
    print(f"{H}x{H} C{C}->N{N} ups{ups}: rel {e:.2e} identical over 12 runs: {same}", flush=True)
print(f"{bad} bad")
sys.exit(1 if bad else 0)
