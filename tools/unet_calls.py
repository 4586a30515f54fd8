"""Per-launcher timing of ONE IR-SDE ConditionalUNet step (B = 8, 256 x 256, bf16) on a single stream: every ops.* call with
its shape, HIP-event milliseconds, achieved TFLOP/s or GB/s.   python tools/unet_calls.py [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes  # noqa: E402
from underwater_image_restoration_amd.unet import ConditionalUNet  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, S = 8, 256
unet = ConditionalUNet(3, 3, 64, [1, 2, 4, 8], 512, True, True, precision=prec)
unet.load_state_dict(synth_state_dict(unet_param_shapes(), seed=2))
unet = unet.cuda().eval()
g = torch.Generator().manual_seed(1)
mu = torch.rand(B, 3, S, S, generator=g).cuda()
xt = mu + torch.randn(B, 3, S, S, generator=g).cuda() * 0.2
tc, ic = torch.randn(B, 512, generator=g).cuda(), torch.randn(B, 512, generator=g).cuda()
for _ in range(3):
    unet(xt, mu, 50, text_context=tc, image_context=ic)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    unet(xt, mu, 50, text_context=tc, image_context=ic)
e1.record()
torch.cuda.synchronize()
print(f"UNet step B{B} {S}x{S} {prec}: {e0.elapsed_time(e1) / 10:.3f} ms")
ops.op_census(True)
unet(xt, mu, 50, text_context=tc, image_context=ic)
torch.cuda.synchronize()
rows = ops.collect_op_census()
ops.op_census(False)
tot = 0.0
for name, a, k, ms in rows:
    tot += ms
    desc = ""
    if name == "conv_gemm":
        A, Wt, out = a[0], a[1], a[2]
        Bn, H, W, C, N = k["B"], k["H"], k["W"], k["C_"], k["N"]
        KH, st, ups, pad = k.get("KH", 1), k.get("stride", 1), k.get("ups", 0), k.get("pad", 0)
        OH = ((H << ups) + 2 * pad - KH) // st + 1
        M, K = Bn * OH * OH, KH * KH * C
        fl = 2.0 * M * N * K
        by = A.numel() // A.shape[1] * C * A.element_size() * (1 if ups == 0 else 1) + (0 if out is None else M * N * out.element_size())
        if k.get("res") is not None:
            by += M * N * 4
        desc = (f"{H}x{W} C{C}->N{N} k{KH} s{st} ups{ups} A:{str(A.dtype)[6:]} out:{str(out.dtype)[6:]} "
                f"{fl / ms / 1e9:7.0f} TF/s  {by / ms / 1e6:6.0f} GB/s")
    elif name in ("ln_rows", "linear_attention", "flash_attn"):
        desc = " ".join(str(tuple(t.shape)) + str(t.dtype)[6:] for t in a if torch.is_tensor(t))[:90]
    print(f"{ms * 1e3:8.1f} us  {name:18s} {desc}")
print(f"sum of calls {tot:.3f} ms")
