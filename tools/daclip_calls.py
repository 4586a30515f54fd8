"""Per-launcher timing of ONE DA-CLIP encode_image(control=True) (B = 16, ViT-B/32, bf16): every ops.* call by shape."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops
from underwater_image_restoration_amd.daclip import DaCLIPVision
from underwater_image_restoration_amd.synth import daclip_vision_param_shapes, synth_state_dict
clip = DaCLIPVision(precision="bf16")
clip.load_state_dict(synth_state_dict(daclip_vision_param_shapes(), seed=1))
clip = clip.cuda().eval()
cin = torch.randn(16, 3, 224, 224, device="cuda")
for _ in range(5):
    clip.encode_image(cin, control=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    clip.encode_image(cin, control=True)
e1.record(); torch.cuda.synchronize()
print(f"encode_image B16: {e0.elapsed_time(e1) / 10:.3f} ms")
ops.op_census(True)
clip.encode_image(cin, control=True)
torch.cuda.synchronize()
rows = ops.collect_op_census()
ops.op_census(False)
agg = collections.OrderedDict()
for name, a, k, ms in rows:
    if name == "conv_gemm":
        key = f"conv_gemm M{a[0].shape[0]} K{k['C_'] * k.get('KH', 1) * k.get('KW', 1)} N{k['N']} A:{str(a[0].dtype)[6:]} out:{str(a[2].dtype)[6:]} " + ",".join(
            x for x in ("bias", "res", "act") if k.get(x) is not None and (torch.is_tensor(k.get(x)) or k.get(x) != 0))
    else:
        key = name + " " + " ".join(str(tuple(t.shape)) for t in a if torch.is_tensor(t))[:70]
    e = agg.setdefault(key, [0, 0.0]); e[0] += 1; e[1] += ms
tot = sum(v[1] for v in agg.values())
for kx, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms * 1e3:8.1f} us {n:4d}x {ms / n * 1e3:7.1f} us  {kx}")
print(f"sum of timed calls {tot:.3f} ms over {sum(v[0] for v in agg.values())} launchers (un-instrumented launchers are not listed)")
