"""A/B/A/B of the student forward (B = 16, 256 x 256, bf16) over uwir_conv_variant values: python tools/exp/abab_variant.py 0,64"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib
from underwater_image_restoration_amd.student import NAFNetWFIGate
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
m = NAFNetWFIGate(**MK, precision="bf16"); m.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0)); m = m.cuda().eval()
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,64").split(",")]
x, z, k = synth_student_inputs(16, 256, 256, 512, 10, mask_hw=(7, 7), seed=1234)
x, z, k = x.cuda(), z.cuda(), k.cuda()
lib = _lib.load()
outs = {}
for rnd in range(3):
    for v in variants:
        lib.uwir_conv_variant(v)
        for _ in range(2): o = m(x, z, k)[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): o = m(x, z, k)[0]
        torch.cuda.synchronize()
        print(f"conv_variant={v}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms", flush=True)
        outs[v] = o
lib.uwir_conv_variant(0)
d = (outs[variants[0]] - outs[variants[-1]]).double()
print("PSNR between the two:", float(10 * torch.log10(1.0 / (d ** 2).mean().clamp_min(1e-20))), "max abs", float(d.abs().max()))
