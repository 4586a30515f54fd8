import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import student_ref as R
from tests.golden.cases import STUDENT_ORACLE_CASES
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
cfg = STUDENT_ORACLE_CASES["prod512"]
mk = dict(base_channels=cfg["base_channels"], num_levels=3, blocks_per_level=cfg["blocks_per_level"], cond_dim=cfg["cond_dim"], num_masks=cfg["num_masks"], num_degradation_types=cfg["num_deg"])
sd = synth_state_dict(student_param_shapes(**mk), seed=cfg["wseed"])
x, z, m = synth_student_inputs(cfg["batch"], cfg["H"], cfg["W"], cfg["cond_dim"], cfg["num_masks"], mask_hw=cfg["mask_hw"], seed=cfg["xseed"])
t0 = time.time()
taps = {}
with torch.no_grad():
    out, _, logits = R.student_forward(sd, x, z, m, blocks_per_level=2, taps=taps)
print("fp32 oracle 512:", time.time() - t0, "s")
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "student_prod512.npz"))
d = np.abs(taps["pre_clamp"].numpy().astype(np.float64) - g["pre_clamp_f64"].astype(np.float64))
print("fp32 oracle vs fp64 oracle: max", d.max(), "p99.99", np.quantile(d, 0.9999), "mean", d.mean())
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "noise512.npy"), np.array([d.max(), np.quantile(d, 0.9999), d.mean()]))
