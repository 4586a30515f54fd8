"""Wall time of the parts of the headline step (DA-CLIP encode, student forward), HIP events + host clock."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.daclip import DaCLIPVision  # noqa: E402
from underwater_image_restoration_amd.student import NAFNetWFIGate  # noqa: E402
from underwater_image_restoration_amd.synth import (daclip_vision_param_shapes, student_param_shapes,  # noqa: E402
                                                    synth_state_dict, synth_student_inputs)


def timeit(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    t_host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, t_host


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mk = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
    model = NAFNetWFIGate(**mk, precision="bf16")
    model.load_state_dict(synth_state_dict(student_param_shapes(**mk), seed=0))
    model = model.cuda().eval()
    clip = DaCLIPVision(precision="bf16")
    clip.load_state_dict(synth_state_dict(daclip_vision_param_shapes(), seed=1))
    clip = clip.cuda().eval()
    x, z, m = synth_student_inputs(B, 256, 256, 512, 10, mask_hw=(7, 7), seed=1234)
    x, z, m = x.cuda(), z.cuda(), m.cuda()
    cin = torch.randn(B, 3, 224, 224).cuda()
    g, h = timeit(lambda: clip.encode_image(cin, control=True))
    print(f"daclip encode_image B{B}: gpu {g:.3f} ms, host enqueue {h:.3f} ms")
    g, h = timeit(lambda: model(x, z, m))
    print(f"student forward B{B}: gpu {g:.3f} ms, host enqueue {h:.3f} ms")
    g, h = timeit(lambda: model(x, clip.encode_image(cin, control=True)[1], m))
    print(f"both B{B}: gpu {g:.3f} ms, host enqueue {h:.3f} ms")


if __name__ == "__main__":
    main()
