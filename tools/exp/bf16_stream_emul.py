"""What would a bf16 residual stream cost in accuracy?  Emulation: every fp32 activation tensor the UNet's kernels write
is rounded to bf16 in place right after the call (worst case: also the SpatialTransformer's internal fp32 tensors).
Compared with the fp32 model: one UNet step (rel L2) and a 100-step reverse_posterior with a fixed noise tape (PSNR)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.sde import IRSDE  # noqa: E402
from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes  # noqa: E402
from underwater_image_restoration_amd.unet import ConditionalUNet  # noqa: E402

EMUL = [False]


def wrap(name, out_of):
    orig = getattr(ops, name)

    def f(*a, **k):
        r = orig(*a, **k)
        if EMUL[0]:
            o = out_of(a, k)
            if o is not None and o.dtype == torch.float32 and o.shape[0] >= 1024:
                o.copy_(o.bfloat16().float())
        return r
    setattr(ops, name, f)


wrap("conv_gemm", lambda a, k: a[2])
wrap("linattn_block", lambda a, k: a[9])
wrap("ln_rows", lambda a, k: a[3])
wrap("add_rows", lambda a, k: a[2])


def psnr(a, b):
    return float(10 * torch.log10(1.0 / ((a.double() - b.double()) ** 2).mean()))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    sd = synth_state_dict(unet_param_shapes(), seed=2)
    nets = {}
    for prec in ("fp32", "bf16"):
        m = ConditionalUNet(3, 3, 64, [1, 2, 4, 8], 512, True, True, precision=prec)
        m.load_state_dict(sd)
        nets[prec] = m.cuda().eval()
    g = torch.Generator().manual_seed(9)
    mu = torch.rand(B, 3, 256, 256, generator=g).cuda()
    tc, ic = torch.randn(B, 512, generator=g).cuda(), torch.randn(B, 512, generator=g).cuda()
    xt = mu + 0.2 * torch.randn(B, 3, 256, 256, generator=g).cuda()
    ref = nets["fp32"](xt, mu, 50, text_context=tc, image_context=ic)
    for emul in (False, True):
        EMUL[0] = emul
        o = nets["bf16"](xt, mu, 50, text_context=tc, image_context=ic)
        print(f"one step, bf16 MFMA, stream {'bf16 (emulated)' if emul else 'fp32'}: rel L2 vs fp32 model "
              f"{float((o - ref).norm() / ref.norm()):.3e}  max abs {float((o - ref).abs().max()):.3e}")
    EMUL[0] = False
    sde = IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device="cuda")
    sde.set_mu(mu)
    x0 = sde.noise_state(mu) if hasattr(sde, "noise_state") else mu + torch.randn(mu.shape, generator=g).cuda() * (50 / 255)
    tape = torch.randn(T, *mu.shape, generator=g).cuda()
    outs = {}
    for tag, prec, emul in (("fp32", "fp32", False), ("bf16", "bf16", False), ("bf16+stream", "bf16", True)):
        EMUL[0] = emul
        sde.set_model(nets[prec])
        outs[tag] = sde.reverse_posterior(x0.clone(), T=T, noise_tape=tape, text_context=tc, image_context=ic).clamp(0, 1)
    EMUL[0] = False
    for tag in ("bf16", "bf16+stream"):
        print(f"{T}-step reverse_posterior, {tag}: PSNR vs the fp32 model's result {psnr(outs[tag], outs['fp32']):.2f} dB, "
              f"max abs {float((outs[tag] - outs['fp32']).abs().max()):.3e}; PSNR(result, mu) fp32 {psnr(outs['fp32'], mu):.3f} "
              f"vs {psnr(outs[tag], mu):.3f} dB")


if __name__ == "__main__":
    main()
