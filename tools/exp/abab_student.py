"""Whole student forward (B = 16, 256^2, bf16; SFFB on its second stream as in bench.py), A/B/A/B over uwir_conv_variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib
from underwater_image_restoration_amd.student import NAFNetWFIGate
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
lib = _lib.load()
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision="bf16")
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
x, z, m = (t.cuda() for t in synth_student_inputs(16, 256, 256, 512, 10, mask_hw=(7, 7), seed=1))
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4]
for _ in range(5):
    model(x, z, m)
torch.cuda.synchronize()
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        lib.uwir_conv_variant(v)
        for _ in range(2):
            model(x, z, m)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            model(x, z, m)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 8)
lib.uwir_conv_variant(0)
for v in variants:
    print(f"variant {v}: " + " ".join(f"{t:.2f}" for t in res[v]) + " ms per forward")
