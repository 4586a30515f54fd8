"""How tight is the a-priori score bound of the bf16 attention fast path?  For every long flash_attn call of one student
forward: max over (b, head) of |c q|_max * max_k |k|  (the bound the kernel uses, per query block in reality), the same
with the keys centred on their mean over the sequence (softmax is invariant to it), and the true max |score|.
    python tools/exp/attn_bound_stats.py [size] [batch]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.student import NAFNetWFIGate  # noqa: E402
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision="bf16")
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
model.overlap_branches = False
x, z, m = synth_student_inputs(B, S, S, 512, 10, mask_hw=(7, 7), seed=99)
orig = ops.flash_attn


def spy(q, k, v1, v2, B_, n, heads, hd, scale, o1, o2, fast=True):
    if n >= 4096:
        c = (scale if scale else 1.0) * 1.4426950408889634
        qf = q.float().reshape(B_, n, -1)[:, :, :heads * hd].reshape(B_, n, heads, hd)
        kf = k.float().reshape(B_, n, -1)[:, :, :heads * hd].reshape(B_, n, heads, hd)
        qn = qf.norm(dim=-1)                       # [B, n, heads]
        kn = kf.norm(dim=-1).amax(dim=1)           # [B, heads]
        kc = (kf - kf.mean(dim=1, keepdim=True)).norm(dim=-1).amax(dim=1)
        ub = c * qn * kn[:, None, :]
        ubc = c * qn * kc[:, None, :]
        # 128*QG-query blocks: a block falls back if ANY of its queries exceeds 90
        blk = 512
        fb = (ub.reshape(B_, n // blk, blk, heads).amax(dim=2) > 90).float().mean().item()
        fbc = (ubc.reshape(B_, n // blk, blk, heads).amax(dim=2) > 90).float().mean().item()
        print(f"n={n} heads={heads} hd={hd} nv={1 if v2 is None else 2}: bound max {ub.max().item():7.1f} median {ub.median().item():6.1f} "
              f"blocks over 90: {100 * fb:5.1f}% | centred keys: max {ubc.max().item():7.1f} median {ubc.median().item():6.1f} "
              f"blocks over 90: {100 * fbc:5.1f}%  (|k| max {kn.max().item():.1f} -> {kc.max().item():.1f}, |q| max {qn.max().item():.1f})")
    return orig(q, k, v1, v2, B_, n, heads, hd, scale, o1, o2, fast=fast)


import underwater_image_restoration_amd.student as st  # noqa: E402
st.ops.flash_attn = spy
with torch.no_grad():
    model(x.cuda(), z.cuda(), m.cuda())
torch.cuda.synchronize()
