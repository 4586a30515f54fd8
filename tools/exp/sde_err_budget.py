"""Where does one bf16 UNet step's error come from?  The fp32 model is run with ONE class of conv_gemm calls emulating bf16
MFMA operands (A and W rounded to bf16 before the exact-fp32 MFMA) or a bf16 output stream (result rounded to bf16), and
the noise prediction is compared with the untouched fp32 model.  Classes = (resolution, kernel size).
    python tools/exp/sde_err_budget.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes  # noqa: E402
from underwater_image_restoration_amd.unet import ConditionalUNet  # noqa: E402

SEL = {"operand": None, "stream": None, "seen": {}}
_orig = ops.conv_gemm


def _cls(k):
    return (k["H"], k.get("KH", 1), k["C_"], k["N"])


def conv_gemm(A, Wt, out, **k):
    c = _cls(k)
    SEL["seen"][c] = SEL["seen"].get(c, 0) + 1
    sel = SEL["operand"]
    if sel is not None and sel(c) and A.dtype == torch.float32:
        A = A.bfloat16().float()
        Wt = Wt.bfloat16().float()
    r = _orig(A, Wt, out, **k)
    sel = SEL["stream"]
    if sel is not None and sel(c) and out.dtype == torch.float32:
        out.copy_(out.bfloat16().float())
    return r


ops.conv_gemm = conv_gemm


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    sd = synth_state_dict(unet_param_shapes(), seed=2)
    net = ConditionalUNet(3, 3, 64, [1, 2, 4, 8], 512, True, True, precision="fp32")
    net.load_state_dict(sd)
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(9)
    mu = torch.rand(B, 3, 256, 256, generator=g).cuda()
    tc, ic = torch.randn(B, 512, generator=g).cuda(), torch.randn(B, 512, generator=g).cuda()
    xt = mu + 0.19 * torch.randn(B, 3, 256, 256, generator=g).cuda()
    ref = net(xt, mu, 50, text_context=tc, image_context=ic).clone()
    classes = sorted(SEL["seen"].items(), key=lambda kv: (-kv[0][0], kv[0][1], kv[0][2]))
    print("classes (H, KH, C, N): calls")
    for c, n in classes:
        print("  ", c, n)

    def rel(o):
        return float((o - ref).norm() / ref.norm())

    def run(kind, sel):
        SEL["operand"], SEL["stream"] = (sel, None) if kind == "operand" else (None, sel)
        o = net(xt, mu, 50, text_context=tc, image_context=ic)
        SEL["operand"] = SEL["stream"] = None
        return rel(o)

    print(f"all conv_gemm operands bf16: {run('operand', lambda c: True):.3e}")
    print(f"all conv_gemm outputs bf16:  {run('stream', lambda c: True):.3e}")
    groups = {
        "stem 7x7": lambda c: c[1] == 7,
        "256^2 3x3": lambda c: c[0] == 256 and c[1] == 3,
        "256^2 1x1": lambda c: c[0] == 256 and c[1] == 1,
        "256^2 4x4 down": lambda c: c[0] == 256 and c[1] == 4,
        "128^2 all": lambda c: c[0] == 128,
        "64^2 all": lambda c: c[0] == 64,
        "32^2 3x3": lambda c: c[0] == 32 and c[1] == 3,
        "32^2 1x1": lambda c: c[0] == 32 and c[1] == 1,
    }
    tot_o = tot_s = 0.0
    for name, sel in groups.items():
        eo, es = run("operand", sel), run("stream", sel)
        tot_o += eo * eo
        tot_s += es * es
        print(f"{name:18s} operands bf16: {eo:.3e}   outputs bf16: {es:.3e}")
    print(f"root sum of squares: operands {tot_o ** 0.5:.3e}, outputs {tot_s ** 0.5:.3e}")
    # single calls at 256^2, in call order
    calls = []
    SEL["seen"] = {}
    idx = [0]

    def by_index(i):
        def sel(c):
            return False
        return sel
    # per-call study at 256^2: count calls in order
    order = []
    orig2 = ops.conv_gemm

    def counting(A, Wt, out, **k):
        order.append(_cls(k))
        return orig2(A, Wt, out, **k)
    ops.conv_gemm = counting
    net(xt, mu, 50, text_context=tc, image_context=ic)
    ops.conv_gemm = orig2
    n_calls = len(order)
    print(f"{n_calls} conv_gemm calls per step; per-call operand rounding at 256^2 and the first / last calls:")
    for target in [i for i, c in enumerate(order) if c[0] == 256]:
        pos = [0]

        def one(A, Wt, out, _t=target, **k):
            i = pos[0]
            pos[0] += 1
            if i == _t and A.dtype == torch.float32:
                A, Wt = A.bfloat16().float(), Wt.bfloat16().float()
            return _orig(A, Wt, out, **k)
        ops.conv_gemm = one
        o = net(xt, mu, 50, text_context=tc, image_context=ic)
        ops.conv_gemm = orig2
        print(f"  call {target:3d} {order[target]}: {rel(o):.3e}")


if __name__ == "__main__":
    main()
