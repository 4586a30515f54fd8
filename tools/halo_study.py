"""Halo-size study for the tiled path: one 512 x 512 frame restored whole (feasible on one GPU) and
through 256-px windows with halo 16 / 32 / 48; PSNR of each tiled result against the whole-frame result, the number of
windows and the recompute factor (window pixels per owned pixel).   python tools/halo_study.py [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.student import NAFNetWFIGate  # noqa: E402
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs  # noqa: E402
from underwater_image_restoration_amd.tiling import TiledRestorer, plan_tiles  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision=prec)
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
# a smooth synthetic scene + texture (pure noise has no structure for a halo to protect)
g = torch.Generator().manual_seed(5)
yy, xx = torch.meshgrid(torch.linspace(0, 1, 512), torch.linspace(0, 1, 512), indexing="ij")
base = torch.stack([0.3 + 0.4 * xx * yy, 0.5 - 0.2 * xx + 0.2 * torch.sin(9 * yy), 0.6 - 0.4 * yy * yy])
x = (base + 0.05 * torch.randn(3, 512, 512, generator=g)).clamp(0, 1)[None].cuda()
_, z, m = synth_student_inputs(1, 8, 8, 512, 10, mask_hw=(7, 7), seed=3)
z, m = z.cuda(), m.cuda()
whole = model(x, z, m)[0]


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 100.0 if mse < 1e-10 else 10.0 * torch.log10(torch.tensor(1.0 / mse)).item()


print(f"precision {prec}; PSNR(whole-frame restored, input) = {psnr(whole, x):.2f} dB (how much the random-weight net changes the image)")
for halo in (0, 16, 32, 48):
    tiles = plan_tiles(512, 512, 256, halo)
    out = TiledRestorer(model, tile=256, halo=halo, max_batch=16)(x, z, m)
    print(f"halo {halo:2d}: {len(tiles):2d} windows, recompute x{len(tiles) * 256 * 256 / 512 ** 2:.2f}, "
          f"PSNR(tiled, whole frame) = {psnr(out, whole):.2f} dB, max abs diff {float((out - whole).abs().max()):.3f}")
