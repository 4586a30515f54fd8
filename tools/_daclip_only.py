import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import ops
from underwater_image_restoration_amd.daclip import DaCLIPVision
from underwater_image_restoration_amd.synth import daclip_vision_param_shapes, synth_state_dict
ops._SPLITK = os.environ.get("SPLITK", "1") == "1"
clip = DaCLIPVision(precision="bf16")
clip.load_state_dict(synth_state_dict(daclip_vision_param_shapes(), seed=1))
clip = clip.cuda().eval()
cin = torch.randn(16, 3, 224, 224, device="cuda")
for _ in range(6):
    clip.encode_image(cin, control=True)
torch.cuda.synchronize()
