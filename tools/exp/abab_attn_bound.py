"""Student forward, A/B/A/B over the a-priori score bound of the packed attention: per-dimension + norm bound (default)
against the norm bound alone (uwir_attn_variant bit 5).   python tools/exp/abab_attn_bound.py [size] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib
from underwater_image_restoration_amd.student import NAFNetWFIGate
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision="bf16")
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
x, z, m = (t.cuda() for t in synth_student_inputs(B, S, S, 512, 10, mask_hw=(7, 7), seed=99))
VARS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8, 40]
outs = {}
for v in VARS:
    lib.uwir_attn_variant(v)
    for _ in range(3):
        out = model(x, z, m)[0]
    outs[v] = out.float().clone()
torch.cuda.synchronize()
d = (outs[VARS[0]] - outs[VARS[-1]]).abs().max().item()
mse = ((outs[VARS[0]] - outs[VARS[-1]]) ** 2).mean().item()
print(f"max |tight - norm-only| = {d:.3e}, PSNR between them {10 * torch.log10(torch.tensor(1.0 / max(mse, 1e-20))).item():.1f} dB")
res = {v: [] for v in VARS}
for rnd in range(3):
    for v in VARS:
        lib.uwir_attn_variant(v)
        model(x, z, m)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            model(x, z, m)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 5)
lib.uwir_attn_variant(-1)
print(f"{S}x{S} B{B} (8: default, 40: norm bound only, 72: 256-query fallback blocks): " + " | ".join(f"variant {v}: " + " ".join(f"{t:.2f}" for t in res[v]) for v in VARS) + " ms per forward")
