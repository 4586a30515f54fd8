"""Micro-benchmark of uwir_flash_attn: python tools/bench_attn.py B n heads hd nv [amp]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops  # noqa: E402


def run(B, n, heads, hd, nv, amp=0.5, fast=True, reps=10, packed=True, lsv=-1):
    from underwater_image_restoration_amd import _lib
    _lib.load().uwir_attn_variant(lsv)
    ops._PACKED = packed
    w = heads * hd
    q, k, v1, v2 = ((torch.randn(B * n, w, device="cuda") * amp).bfloat16() for _ in range(4))
    o1 = torch.empty(B * n, w, device="cuda", dtype=torch.bfloat16)
    o2 = torch.empty_like(o1) if nv == 2 else None
    f = lambda: ops.flash_attn(q, k, v1, v2 if nv == 2 else None, B, n, heads, hd, 1 / math.sqrt(hd), o1, o2, fast=fast)  # noqa: E731
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = (2 + 2 * nv) * B * heads * float(n) * n * hd
    units = B * heads * (n / 32) ** 2 / 1024
    print(f"B{B} n{n} h{heads} hd{hd} nv{nv} amp{amp} fast={fast} packed={packed} lsv={lsv}: {ms:.3f} ms {fl / ms / 1e9:6.0f} TF/s "
          f"{ms * 1e-3 * 2.4e9 / units:5.0f} cyc/unit")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lsv":
        for cfg in ((16, 16384, 12, 16, 1), (16, 16384, 4, 16, 2), (16, 4096, 12, 32, 1), (16, 4096, 4, 32, 2)):
            for lsv in (0, 1, 2, 4, 5, 6):
                run(*cfg, fast=True, lsv=lsv)
    elif len(sys.argv) > 5:
        a = [int(v) for v in sys.argv[1:6]]
        amp = float(sys.argv[6]) if len(sys.argv) > 6 else 0.5
        run(*a, amp=amp, fast=True)
        run(*a, amp=amp, fast=False)
    else:
        for cfg in ((16, 16384, 12, 16, 1), (16, 16384, 4, 16, 2), (16, 4096, 12, 32, 1), (16, 4096, 4, 32, 2),
                    (16, 1024, 12, 64, 1), (16, 1024, 4, 64, 2), (16, 50, 12, 64, 1), (4, 65536, 12, 16, 1)):
            run(*cfg, fast=True)
            run(*cfg, fast=True, packed=False)
            run(*cfg, fast=False)
