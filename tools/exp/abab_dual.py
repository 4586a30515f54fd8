"""Student forward (B = 16, 256^2, bf16): single forward vs pipeline.DualReplicaRestorer, each with and without the
attention occupancy cap (uwir_attn_variant bit 4), A/B/A/B.   python tools/exp/abab_dual.py [stagger_after]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import _lib
from underwater_image_restoration_amd.pipeline import DualReplicaRestorer
from underwater_image_restoration_amd.student import NAFNetWFIGate
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
lib = _lib.load()
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision="bf16")
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
x, z, m = (t.cuda() for t in synth_student_inputs(16, 256, 256, 512, 10, mask_hw=(7, 7), seed=1))
stag = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dual = DualReplicaRestorer(model, stagger_after=stag)


def run_single(k):
    for _ in range(k):
        model(x, z, m)


def run_dual(k):
    hs = [dual.submit(x, z, m) for _ in range(k)]
    for h in hs:
        dual.wait(h)


cases = [("single", 8, run_single), ("single+cap", 24, run_single), ("dual", 8, run_dual), ("dual+cap", 24, run_dual)]
for _, v, fn in cases:
    lib.uwir_attn_variant(v)
    fn(3)
torch.cuda.synchronize()
res = {c[0]: [] for c in cases}
for rnd in range(3):
    for name, v, fn in cases:
        lib.uwir_attn_variant(v)
        fn(2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(8)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 8)
lib.uwir_attn_variant(-1)
for name, _, _ in cases:
    print(f"{name:12s}: " + " ".join(f"{t:.2f}" for t in res[name]) + " ms per forward", flush=True)
