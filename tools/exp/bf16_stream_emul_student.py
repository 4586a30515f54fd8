"""What would a bf16 activation stream cost the STUDENT in accuracy?  Emulation: after every kernel call, every large fp32
row matrix among its arguments is rounded to bf16 in place (SFFB amplitude / phase internals and fp32-compute GEMMs
excepted).  Compared against the fp32 model on the benchmark shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.student import NAFNetWFIGate  # noqa: E402
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs  # noqa: E402

EMUL = [False]
SKIP = {"fft2_amp_phase", "fft_workspace_bytes"}


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        r = orig(*a, **k)
        if EMUL[0] and name not in SKIP:
            ts = [t for t in list(a) + list(k.values()) if torch.is_tensor(t)]
            if name == "ifft2_polar_add":
                ts = [a[7]]
            if name == "conv_gemm" and a[1].dtype == torch.float32:
                ts = []
            for t in ts:
                if t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] >= 4096:
                    t.copy_(t.bfloat16().float())
        return r
    setattr(ops, name, f)


for n in ("conv_gemm", "dwt2", "idwt2", "dwconv_ms", "flash_attn", "ifft2_polar_add", "fft2_amp_phase", "affine_cast",
          "ln2d_affine", "bilinear", "stem_conv3x3", "head_conv3x3"):
    if hasattr(ops, n):
        wrap(n)


def psnr(a, b):
    return float(10 * torch.log10(1.0 / ((a.double() - b.double()) ** 2).mean()))


MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
sd = synth_state_dict(student_param_shapes(**MK), seed=0)
x, z, m = (t.cuda() for t in synth_student_inputs(2, 256, 256, 512, 10, mask_hw=(7, 7), seed=1))
outs = {}
for prec in ("fp32", "bf16"):
    model = NAFNetWFIGate(**MK, precision=prec)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    outs[prec] = model(x, z, m)[0].clone()
    if prec == "bf16":
        EMUL[0] = True
        outs["bf16+stream"] = model(x, z, m)[0].clone()
        EMUL[0] = False
for k in ("bf16", "bf16+stream"):
    print(f"{k}: PSNR vs the fp32 model {psnr(outs[k], outs['fp32']):.2f} dB, max abs {float((outs[k] - outs['fp32']).abs().max()):.3e}")
