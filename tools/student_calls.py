"""Per-launcher timing of ONE student forward (B = 16, 256 x 256) on a single stream, aggregated by (launcher, shape):
calls, total ms, achieved TFLOP/s and GB/s.   python tools/student_calls.py [precision] [batch] [size]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402
from underwater_image_restoration_amd.student import NAFNetWFIGate  # noqa: E402
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
MK = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
model = NAFNetWFIGate(**MK, precision=prec)
model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
model = model.cuda().eval()
model.overlap_branches = False
x, z, m = synth_student_inputs(B, S, S, 512, 10, mask_hw=(7, 7), seed=1234)
x, z, m = x.cuda(), z.cuda(), m.cuda()
for _ in range(3):
    model(x, z, m)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    model(x, z, m)
e1.record()
torch.cuda.synchronize()
print(f"student forward B{B} {S}x{S} {prec} (single stream): {e0.elapsed_time(e1) / 5:.3f} ms")
ops.op_census(True)
model(x, z, m)
torch.cuda.synchronize()
rows = ops.collect_op_census()
ops.op_census(False)
agg = OrderedDict()
tot = 0.0
for name, a, k, ms in rows:
    tot += ms
    fl = by = 0.0
    if name == "conv_gemm":
        A, Wt, out = a[0], a[1], a[2]
        Bn, H, W, C, N = k["B"], k["H"], k["W"], k["C_"], k["N"]
        KH, KW, st, ups, pad = k.get("KH", 1), k.get("KW", 1), k.get("stride", 1), k.get("ups", 0), k.get("pad", 0)
        OH, OW = ((H << ups) + 2 * pad - KH) // st + 1, ((W << ups) + 2 * pad - KW) // st + 1
        M, K = Bn * OH * OW, KH * KW * C
        fl = 2.0 * M * N * K
        by = Bn * H * W * C * A.element_size() + (0 if out is None else M * N * out.element_size())
        res = k.get("res")
        if res is not None:
            by += M * N * res.element_size()
        extra = ",".join(f"{kk}" for kk in ("a_scale", "bias", "batch_bias", "res", "res_scale", "col_scale", "batch_scale", "blend_out") if k.get(kk) is not None)
        key = (f"conv_gemm M{M} N{N} K{K} k{KH} s{st} A:{str(A.dtype)[6:]} out:{'-' if out is None else str(out.dtype)[6:]} "
               f"act{k.get('act', 0)} sm{k.get('store_mode', 0)} [{extra}]")
    else:
        shp = " ".join(str(tuple(t.shape)) + str(t.dtype)[6:] for t in a if torch.is_tensor(t))[:100]
        key = f"{name} {shp}"
        for t in a:
            if torch.is_tensor(t):
                by += t.numel() * t.element_size()
    e = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    e[0] += 1
    e[1] += ms
    e[2] += fl
    e[3] += by
for key, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.3f} ms {n:4d}x {ms / n * 1e3:8.1f} us  {fl / ms / 1e9 if fl else 0:7.0f} TF/s {by / ms / 1e6:7.0f} GB/s(args)  {key}")
print(f"sum of calls {tot:.3f} ms")
