"""Micro-benchmark of uwir_fft2_amp_phase / uwir_ifft2_polar_add at the student's three SFFB shapes (B 16, fp32), radix-4
line kernel against the radix-2 one (uwir_conv_variant bit 2), with the difference between their results.
    python tools/bench_fft.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import _lib, ops  # noqa: E402

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = ((16, 128, 128, 64), (16, 64, 64, 128), (16, 32, 32, 256))
if len(sys.argv) > 2 and sys.argv[2] == "mixed":   # sub-band maps of a 240 x 320 image: 2^a 3^b 5^c lines (old form: direct DFT)
    shapes = ((16, 120, 160, 64), (16, 60, 80, 128), (16, 30, 40, 256), (1, 270, 480, 64))
for B, H, W, C in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * H * W, C, generator=g).cuda()
    add = torch.randn(B * H * W, C, generator=g).cuda()
    ws = torch.empty(ops.fft_workspace_bytes(B, H, W, C) // 4, device="cuda")
    res = {}
    for variant in (4, 0):
        lib.uwir_conv_variant(variant)
        amp, ph, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        f = lambda: ops.fft2_amp_phase(x, B, H, W, C, amp, ph, ws)            # noqa: E731
        i = lambda: ops.ifft2_polar_add(amp, ph, add, B, H, W, C, out, ws)    # noqa: E731
        for _ in range(3):
            f(); i()
        torch.cuda.synchronize()
        ts = []
        for fn in (f, i):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        res[variant] = (amp.clone(), out.clone())
        by_f, by_i = x.numel() * 4 * 3, x.numel() * 4 * 4
        print(f"{'radix-2 / direct DFT' if variant else 'radix-4 / mixed radix'} B{B} {H}x{W} C{C}: fft2 {ts[0]*1e3:7.1f} us ({by_f/ts[0]/1e6:5.0f} GB/s algorithmic)  "
              f"ifft2 {ts[1]*1e3:7.1f} us ({by_i/ts[1]/1e6:5.0f} GB/s)")
    lib.uwir_conv_variant(0)
    # x -> fft -> ifft (+ addend) must give x + addend back
    print(f"   amp max diff {float((res[4][0]-res[0][0]).abs().max()):.2e} (|amp| max {float(res[0][0].abs().max()):.1f}), "
          f"round trip error {float((res[0][1] - x - add).abs().max()):.2e}")
