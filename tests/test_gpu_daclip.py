"""GPU parity of the DA-CLIP image encoder (two ViT towers on HIP kernels through the C ABI) against the CPU oracle
and golden vectors produced by the reference's own open_clip classes."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import daclip_ref as D
from tests.golden.cases import DACLIP_CASES, daclip_image
from tests.helpers import GOLDEN, max_abs
from underwater_image_restoration_amd.synth import daclip_vision_param_shapes, synth_state_dict

pytestmark = pytest.mark.gpu


def _setup(name, precision):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.daclip import DaCLIPVision
    cfg = DACLIP_CASES[name]
    sd = synth_state_dict(daclip_vision_param_shapes(cfg["image_size"], cfg["patch_size"], cfg["width"], cfg["layers"],
                                                     cfg["output_dim"]), seed=cfg["wseed"])
    m = DaCLIPVision(cfg["image_size"], cfg["patch_size"], cfg["width"], cfg["layers"], cfg["heads"], cfg["output_dim"],
                     precision=precision)
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return cfg, sd, m.cuda().eval(), daclip_image(cfg), np.load(os.path.join(GOLDEN, f"daclip_{name}.npz"))


@pytest.mark.parametrize("name", ["small", "grid7", "vitb32"])
def test_encode_image_fp32(name):
    cfg, sd, model, img, g = _setup(name, "fp32")
    image, degra = model.encode_image(img.cuda(), control=True)
    e_i, e_d = max_abs(image.cpu().numpy(), g["image_f64"]), max_abs(degra.cpu().numpy(), g["degra_f64"])
    print(f"\n[daclip {name}/fp32] image err {e_i:.3e} degra err {e_d:.3e} (|ref| ~ {np.abs(g['image_f64']).max():.2f})")
    assert image.shape == (cfg["batch"], cfg["output_dim"]) and image.dtype == torch.float32
    assert e_i < 1e-4 and e_d < 1e-4                                  # vs the reference run in fp64
    assert max_abs(image.cpu().numpy(), g["image"]) < 2e-4            # vs the reference in fp32
    plain = model.encode_image(img.cuda(), control=False)
    assert max_abs(plain.cpu().numpy(), g["plain"]) < 2e-4
    im2, dg2 = model.encode_image(img.cuda(), control=True)           # cached buffers, deterministic kernels
    assert torch.equal(image, im2) and torch.equal(degra, dg2)
    ni, nd = model.encode_image(img.cuda(), control=True, normalize=True)
    assert max_abs(ni.cpu().numpy(), F.normalize(torch.from_numpy(g["image_f64"]), dim=-1).numpy()) < 1e-4
    assert max_abs(nd.cpu().numpy(), F.normalize(torch.from_numpy(g["degra_f64"]), dim=-1).numpy()) < 1e-4


@pytest.mark.parametrize("name", ["small", "grid7"])
def test_tower_calls_and_hiddens(name):
    """visual_control(x, output_hiddens=True) / visual(x, control=hiddens) as separate calls (export_pseudolabels.py:274)."""
    cfg, sd, model, img, g = _setup(name, "fp32")
    degra, hiddens = model.visual_control(img.cuda(), output_hiddens=True)
    assert len(hiddens) == cfg["layers"]
    L = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
    for i, h in enumerate(hiddens):
        assert tuple(h.shape) == (L, cfg["batch"], cfg["width"])      # LND like the reference
        assert max_abs(h.permute(1, 0, 2).cpu().numpy(), g[f"hidden{i}"]) < 2e-4, i
    image = model.visual(img.cuda(), control=hiddens)
    assert len(hiddens) == 0                                          # consumed, like control.pop()
    assert max_abs(image.cpu().numpy(), g["image_f64"]) < 1e-4
    assert max_abs(degra.cpu().numpy(), g["degra_f64"]) < 1e-4
    assert max_abs(model.visual(img.cuda()).cpu().numpy(), g["plain"]) < 2e-4
    assert max_abs(model.visual_control(img.cuda()).cpu().numpy(), g["degra_f64"]) < 1e-4


@pytest.mark.parametrize("name", ["grid7", "vitb32"])
def test_encode_image_bf16(name):
    cfg, sd, model, img, g = _setup(name, "bf16")
    image, degra = model.encode_image(img.cuda(), control=True)
    for got, key in ((image, "image_f64"), (degra, "degra_f64")):
        ref = torch.from_numpy(g[key])
        cos = F.cosine_similarity(got.cpu().double(), ref, dim=-1).min()
        rel = float((got.cpu().double() - ref).norm() / ref.norm())
        print(f"\n[daclip {name}/bf16] {key}: min cosine {float(cos):.6f}  rel L2 err {rel:.3e}")
        assert cos > 0.999 and rel < 3e-2


def test_api_errors():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.daclip import DaCLIPVision
    m = DaCLIPVision(64, 32, 128, 1, 2, 32).cuda()
    with pytest.raises(ValueError):
        m.encode_image(torch.zeros(1, 3, 32, 32, device="cuda"))
    with pytest.raises(RuntimeError):
        m.encode_image(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ValueError):
        DaCLIPVision(64, 32, 100, 1, 3, 32)
    # default init: zero control links => control=True leaves the image features unchanged (transformer.py:295-306)
    x = torch.randn(2, 3, 64, 64, device="cuda")
    a, _ = m.encode_image(x, control=True)
    assert torch.allclose(a, m.encode_image(x, control=False), atol=1e-6)
