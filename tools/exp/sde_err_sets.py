"""Which tensors / calls of a UNet step have to stay above bf16 for the step's error to drop by a given factor?  The fp32
model is run with EVERYTHING emulating the bf16 mode (operands of every conv_gemm rounded to bf16, every activation the
kernels write rounded to bf16) except a chosen set of conv_gemm calls (operands exact) and outputs (not rounded).
    python tools/exp/sde_err_sets.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes  # noqa: E402
from underwater_image_restoration_amd.unet import ConditionalUNet  # noqa: E402

ST = {"on": False, "exact": set(), "keep": set(), "ci": -1, "log": []}
_conv = ops.conv_gemm


def rnd(t):
    return t.bfloat16().float()


def conv_gemm(A, Wt, out, **k):
    ST["ci"] += 1
    ci = ST["ci"]
    ST["log"].append(("conv", ci, k["H"], k.get("KH", 1), k["C_"], k["N"]))
    if ST["on"] and ci not in ST["exact"] and A.dtype == torch.float32:
        A, Wt = rnd(A), rnd(Wt)
    r = _conv(A, Wt, out, **k)
    if ST["on"] and ("conv", ci) not in ST["keep"] and out.dtype == torch.float32 and out.shape[0] >= 1024:
        out.copy_(rnd(out))
    return r


def wrap(name, out_of):
    orig = getattr(ops, name)

    def f(*a, **k):
        r = orig(*a, **k)
        key = (name, ST["ci"])
        ST["log"].append(key)
        o = out_of(a, k)
        if ST["on"] and key not in ST["keep"] and o is not None and o.dtype == torch.float32 and o.shape[0] >= 1024:
            o.copy_(rnd(o))
        return r
    setattr(ops, name, f)


ops.conv_gemm = conv_gemm
wrap("ln_rows", lambda a, k: a[3])
wrap("add_rows", lambda a, k: a[2])
wrap("linear_attention", lambda a, k: a[4])


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

    def run(exact=(), keep=()):
        ST.update(on=True, exact=set(exact), keep=set(keep), ci=-1, log=[])
        o = net(xt, mu, 50, text_context=tc, image_context=ic).clone()
        ST["on"] = False
        return o

    ST.update(on=False, ci=-1, log=[])
    ref = net(xt, mu, 50, text_context=tc, image_context=ic).clone()
    print("call sequence after conv 72:")
    for e in ST["log"]:
        if (e[0] == "conv" and e[1] >= 72) or (e[0] != "conv" and e[1] >= 72):
            print("   ", e)

    def rel(o):
        return float((o - ref).norm() / ref.norm())

    print(f"everything bf16 (emulated): {rel(run()):.3e}")
    c = lambda i: ("conv", i)  # noqa: E731
    sets = [
        ("conv 84 exact (catF halves = outputs of 0, 82 and its own output kept)", [84], [c(0), c(82), c(84)]),
        ("+ fin kept (output of 85)", [84], [c(0), c(82), c(84), c(85)]),
        ("+ conv 82 exact, its input kept", [84, 82], [c(0), c(82), c(84), c(85), ("ln_rows", 81), ("add_rows", 81)]),
        ("+ stem (conv 0) exact", [84, 82, 0], [c(0), c(82), c(84), c(85), ("ln_rows", 81), ("add_rows", 81)]),
        ("+ convs 83, 85 exact, h kept", [84, 82, 0, 83, 85], [c(0), c(82), c(83), c(84), c(85), ("ln_rows", 81), ("add_rows", 81)]),
        ("only stem + 82 + 84 + 83 + 85 exact, nothing kept", [84, 82, 0, 83, 85], []),
        ("nothing exact; catF / res / fin / t3 kept", [], [c(0), c(82), c(84), c(85), ("ln_rows", 81), ("add_rows", 81)]),
    ]
    for name, ex, keep in sets:
        print(f"{name:75s}: {rel(run(ex, keep)):.3e}")


if __name__ == "__main__":
    main()
