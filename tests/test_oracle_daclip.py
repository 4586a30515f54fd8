"""CPU: pin oracle/daclip_ref.py against golden vectors produced by the reference's own open_clip classes
(tests/golden/make_golden.py::daclip_case)."""
import os

import numpy as np
import pytest
import torch

from oracle import daclip_ref as D
from tests.golden.cases import DACLIP_CASES, daclip_image
from tests.helpers import max_abs
from underwater_image_restoration_amd.synth import daclip_vision_param_shapes, synth_state_dict


def _setup(name, golden_dir):
    cfg = DACLIP_CASES[name]
    sd = synth_state_dict(daclip_vision_param_shapes(cfg["image_size"], cfg["patch_size"], cfg["width"], cfg["layers"],
                                                     cfg["output_dim"]), seed=cfg["wseed"])
    return cfg, sd, daclip_image(cfg), np.load(os.path.join(golden_dir, f"daclip_{name}.npz"))


@pytest.mark.parametrize("name", ["small", "grid7"])
def test_daclip_oracle_matches_reference(golden_dir, name):
    cfg, sd, img, g = _setup(name, golden_dir)
    with torch.no_grad():
        image, degra = D.encode_image(sd, img, cfg["heads"], control=True)
        _, hiddens = D.control_tower(sd, img, cfg["heads"])
        plain = D.encode_image(sd, img, cfg["heads"], control=False)
    assert max_abs(image.numpy(), g["image"]) < 2e-5 and max_abs(degra.numpy(), g["degra"]) < 2e-5
    assert max_abs(plain.numpy(), g["plain"]) < 2e-5
    assert len(hiddens) == cfg["layers"]
    for i, h in enumerate(hiddens):
        assert max_abs(h.numpy(), g[f"hidden{i}"]) < 2e-5, i
    # fp64: same math to ~1e-12 (independent of fp32 summation order)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        i64, d64 = D.encode_image(sd64, img.double(), cfg["heads"], control=True)
    assert max_abs(i64.numpy(), g["image_f64"]) < 1e-10 and max_abs(d64.numpy(), g["degra_f64"]) < 1e-10
    # control wiring really matters in these fixtures (zero-modules are randomised)
    assert max_abs(plain.numpy(), g["image"]) > 1e-2


def test_daclip_oracle_vitb32(golden_dir):
    cfg, sd, img, g = _setup("vitb32", golden_dir)
    with torch.no_grad():
        image, degra = D.encode_image(sd, img, cfg["heads"], control=True)
    assert image.shape == (1, 512) and degra.shape == (1, 512)
    assert max_abs(image.numpy(), g["image"]) < 5e-5 and max_abs(degra.numpy(), g["degra"]) < 5e-5


def test_normalize_flag():
    cfg = DACLIP_CASES["small"]
    sd = synth_state_dict(daclip_vision_param_shapes(cfg["image_size"], cfg["patch_size"], cfg["width"], cfg["layers"],
                                                     cfg["output_dim"]), seed=cfg["wseed"])
    with torch.no_grad():
        image, degra = D.encode_image(sd, daclip_image(cfg), cfg["heads"], control=True, normalize=True)
    assert torch.allclose(image.norm(dim=-1), torch.ones(3), atol=1e-6)
    assert torch.allclose(degra.norm(dim=-1), torch.ones(3), atol=1e-6)
