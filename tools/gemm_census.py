"""Per-shape census of the 1x1 / conv GEMM calls of one student step (B=16, 256x256, bf16): for each distinct
(M, N, K, A dtype, out dtype, epilogue) the call count, time, TFLOP/s and algorithmic GB/s (A + out + residual/blend
streams).   python tools/gemm_census.py [batch] [serial]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.student import NAFNetWFIGate  # noqa: E402
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision="bf16")
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
if "serial" in sys.argv:      # SFFB on the main stream: a call's time is its own
    model.overlap_branches = False
x, z, m = synth_student_inputs(B, 256, 256, 512, 10, mask_hw=(7, 7), seed=1)
x, z, m = x.cuda(), z.cuda(), m.cuda()
for _ in range(2):
    model(x, z, m)
torch.cuda.synchronize()
ops.op_census(True)
model(x, z, m)
torch.cuda.synchronize()
calls = ops.collect_op_census()
ops.op_census(False)

es = lambda t: 0 if t is None else t.element_size()
rows = OrderedDict()
tot = 0.0
for name, a, k, ms in calls:
    if name != "conv_gemm":
        continue
    A, Wt, out = a[0], a[1], a[2]
    Bn, H, W, C_, N = k["B"], k["H"], k["W"], k["C_"], k["N"]
    KH, KW, stride, ups = k.get("KH", 1), k.get("KW", 1), k.get("stride", 1), k.get("ups", 0)
    OH = k.get("OH") or ((H << ups) + 2 * k.get("pad", 0) - KH) // stride + 1
    OW = k.get("OW") or ((W << ups) + 2 * k.get("pad", 0) - KW) // stride + 1
    M, K = Bn * OH * OW, KH * KW * C_
    epi = [n for n in ("a_scale", "bias", "batch_bias", "res", "res_scale", "blend_t", "blend_out", "col_scale", "batch_scale")
           if k.get(n) is not None]
    if k.get("act"):
        epi.append(f"act{k['act']}")
    if k.get("store_mode"):
        epi.append(f"store{k['store_mode']}")
    byts = Bn * H * W * C_ * es(A) + M * N * es(out) + N * K * es(Wt)
    if k.get("res") is not None:
        byts += M * N * 4
    if k.get("blend_t") is not None:
        byts += 2 * M * N * 4
    if k.get("blend_out") is not None:
        byts += M * N * 4
    key = (M, N, K, str(A.dtype)[6:], "-" if out is None else str(out.dtype)[6:], ",".join(epi))
    r = rows.setdefault(key, [0, 0.0, 2.0 * M * N * K, byts])
    r[0] += 1
    r[1] += ms
    tot += ms
print(f"student step B={B}: conv_gemm total {tot:.3f} ms")
print(f"{'M':>8} {'N':>5} {'K':>5} {'A':>8} {'out':>8} {'calls':>5} {'ms/call':>8} {'TF/s':>7} {'GB/s':>7}  epilogue")
for (M, N, K, ad, od, epi), (n, ms, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    print(f"{M:8d} {N:5d} {K:5d} {ad:>8} {od:>8} {n:5d} {per:8.4f} {fl / per / 1e9:7.1f} {by / per / 1e6:7.0f}  {epi}")
