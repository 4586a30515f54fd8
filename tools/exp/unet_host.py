"""Host enqueue time against GPU time of one ConditionalUNet step (B = 8, 256 x 256, bf16): is the sampler host-bound?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes
from underwater_image_restoration_amd.unet import ConditionalUNet
B, S = 8, 256
unet = ConditionalUNet(3, 3, 64, [1, 2, 4, 8], 512, True, True, precision="bf16")
unet.load_state_dict(synth_state_dict(unet_param_shapes(), seed=2))
unet = unet.cuda().eval()
g = torch.Generator().manual_seed(1)
mu = torch.rand(B, 3, S, S, generator=g).cuda()
xt = mu + torch.randn(B, 3, S, S, generator=g).cuda() * 0.2
tc, ic = torch.randn(B, 512, generator=g).cuda(), torch.randn(B, 512, generator=g).cuda()
for _ in range(5):
    unet(xt, mu, 50, text_context=tc, image_context=ic)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        unet(xt, mu, 50, text_context=tc, image_context=ic)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms / step, total {(t2 - t0) / 20 * 1e3:.3f} ms / step")
with unet.conditioning_schedule(list(range(50, 30, -1)), B, tc, ic):
    t0 = time.perf_counter()
    for t in range(50, 30, -1):
        unet(xt, mu, t, text_context=tc, image_context=ic)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"scheduled: host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms / step, total {(t2 - t0) / 20 * 1e3:.3f} ms / step")
