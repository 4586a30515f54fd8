"""flash_attn with scores of a common sign and size (q ~ a u, k ~ s b u: every score of a query near s * target log2 units),
over head sizes, value-stream counts, ragged and whole sequences, packed and LDS shapes; against fp64.
    python tools/fuzz_attn_offsets.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(11)
bad = tot = 0
for hd in (16, 32, 64):
    for nv in (1, 2):
        for (b, n, heads) in ((1, 2048 + 13, 32), (1, 2048, 32), (2, 300, 4), (1, 1024 + 31, 64), (1, 777, 12)):
            for target in (-85.0, -50.0, -20.0, 20.0, 50.0, 85.0):
                w = heads * hd
                c = 1.4426950408889634 / math.sqrt(hd)
                amp = math.sqrt(abs(target) / c)
                u = torch.nn.functional.normalize(torch.randn(heads, hd, generator=g), dim=-1)
                q = (u[None] * amp + 0.05 * torch.randn(b * n, heads, hd, generator=g)).reshape(b * n, w).bfloat16().cuda()
                k = (math.copysign(1.0, target) * u[None] * amp * 0.97 + 0.05 * torch.randn(b * n, heads, hd, generator=g)).reshape(b * n, w).bfloat16().cuda()
                v1 = torch.randn(b * n, w, generator=g).bfloat16().cuda()
                v2 = torch.randn(b * n, w, generator=g).bfloat16().cuda() if nv == 2 else None
                o = [torch.full((b * n, w), float("nan"), device="cuda") for _ in range(nv)]
                ops.flash_attn(q, k, v1, v2, b, n, heads, hd, 1 / math.sqrt(hd), o[0], o[1] if nv == 2 else None)
                torch.cuda.synchronize()
                for oi, v in zip(o, [v1, v2][:nv]):
                    tot += 1
                    worst = 0.0
                    for bi in range(b):
                        rows = slice(bi * n, (bi + 1) * n)
                        for h in range(0, heads, max(heads // 4, 1)):
                            cols = slice(h * hd, (h + 1) * hd)
                            a = torch.softmax(q[rows, cols].double() @ k[rows, cols].double().T / math.sqrt(hd), dim=-1)
                            ref = a @ v[rows, cols].double()
                            got = oi[rows, cols].double()
                            e = float((got - ref).norm() / ref.norm()) if bool(torch.isfinite(got).all()) else float("inf")
                            worst = max(worst, e)
                    if not worst < 3e-2:
                        bad += 1
                        print(f"MISMATCH hd{hd} nv{nv} b{b} n{n} heads{heads} target {target}: rel {worst:.3e}", flush=True)
print(f"attention offsets: {tot} outputs, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
