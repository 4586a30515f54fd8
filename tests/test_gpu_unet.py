"""GPU parity of the IR-SDE ConditionalUNet (HIP path through the C ABI) against the CPU oracle and the golden
vectors produced by the reference module itself, plus the full reverse-SDE loop with the real score network."""
import os

import numpy as np
import pytest
import torch

from oracle import sde_ref as S
from oracle import unet_ref as U
from tests.golden.cases import UNET_CASES, unet_inputs
from tests.helpers import GOLDEN, max_abs, psnr
from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-4  # north_star: fp32 max abs error <= 1e-4


def _setup(name, precision):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.unet import ConditionalUNet
    cfg = UNET_CASES[name]
    sd = synth_state_dict(unet_param_shapes(3, 3, cfg["nf"], cfg["ch_mult"], 512, cfg["degra"], cfg["image"]), seed=cfg["wseed"])
    m = ConditionalUNet(3, 3, cfg["nf"], cfg["ch_mult"], 512, cfg["degra"], cfg["image"], precision=precision)
    r = m.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)  # DataParallel-style prefix
    assert not r.missing_keys and not r.unexpected_keys
    return cfg, sd, m.cuda().eval(), unet_inputs(cfg), np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))


def _oracle64(sd, xt, mu, t, text, image, taps=None):
    d = lambda z: None if z is None else z.double()  # noqa: E731
    U.FORCE_EPS = 1e-5  # high-precision evaluation of the fp32 function (the reference's fp64 branch uses 1e-3)
    try:
        with torch.no_grad():
            return U.unet_forward({k: v.double() for k, v in sd.items()}, xt.double(), mu.double(),
                                  torch.tensor([float(t)], dtype=torch.float64), d(text), d(image), taps=taps)
    finally:
        U.FORCE_EPS = None


def _c(z):
    return None if z is None else z.cuda()


@pytest.mark.parametrize("name", list(UNET_CASES))
def test_unet_fp32_matches_oracle_and_golden(name):
    cfg, sd, model, (xt, mu, text, image), g = _setup(name, "fp32")
    for ti, t in enumerate(cfg["times"]):
        rt, gt = {}, {}
        ref = _oracle64(sd, xt, mu, t, text, image, taps=rt)
        out = model(xt.cuda(), mu.cuda(), t, text_context=_c(text), image_context=_c(image), taps=gt)
        torch.cuda.synchronize()
        rows = [(k, float((gt[k].double().cpu() - rt[k]).abs().max()), float(rt[k].abs().max()))
                for k in gt if k in rt and k != "st"]
        print(f"\n[unet {name}/fp32 t={t}] " + "  ".join(f"{k}:{e:.1e}({m:.1f})" for k, e, m in rows))
        e = max_abs(out.cpu().numpy(), ref.numpy())
        print(f"[unet {name}/fp32 t={t}] out err {e:.3e}  vs golden(ref fp32) {max_abs(out.cpu().numpy(), g[f'out{ti}']):.3e}")
        assert out.shape == xt.shape and out.dtype == torch.float32
        assert e < FP32_TOL
        assert max_abs(out.cpu().numpy(), g[f"out{ti}"]) < 2 * FP32_TOL
        out2 = model(xt.cuda(), mu.cuda(), t, text_context=_c(text), image_context=_c(image))
        assert torch.equal(out, out2)


@pytest.mark.parametrize("name", ["full4", "prod"])
def test_unet_bf16(name):
    cfg, sd, model, (xt, mu, text, image), g = _setup(name, "bf16")
    t = cfg["times"][0]
    ref = _oracle64(sd, xt, mu, t, text, image)
    out = model(xt.cuda(), mu.cuda(), t, text_context=_c(text), image_context=_c(image)).cpu()
    rel = float((out.double() - ref).norm() / ref.norm())
    print(f"\n[unet {name}/bf16] rel L2 err {rel:.3e}  max abs {float((out.double() - ref).abs().max()):.3e}")
    assert rel < 3e-2


def test_reverse_sampling_with_the_real_score_network():
    """Config 4 in miniature: IRSDE.reverse_sde / reverse_posterior driving the HIP UNet, vs the oracle loop."""
    cfg, sd, model, (xt, mu, text, image), g = _setup("lin", "fp32")
    from underwater_image_restoration_amd.sde import IRSDE
    T = 5
    gen = torch.Generator().manual_seed(17)
    tape = torch.randn(T, *xt.shape, generator=gen)
    sde = IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device="cuda")
    sde.set_mu(mu.cuda())
    sde.set_model(model)
    tb = S.SdeTables(50, T=100, schedule="cosine", eps=0.005)
    U.FORCE_EPS = 1e-5
    try:
        sd64 = {k: v.double() for k, v in sd.items()}
        net = lambda x, m, t: U.unet_forward(sd64, x, m, torch.tensor([float(t)], dtype=torch.float64), text.double(), None)  # noqa: E731
        for mode in ("sde", "posterior"):
            with torch.no_grad():
                ref = tb.sample(net, xt.double(), mu.double(), mode, tape.double(), T=T)
            fn = sde.reverse_sde if mode == "sde" else sde.reverse_posterior
            out = fn(xt.cuda(), T=T, noise_tape=tape.cuda(), text_context=text.cuda())
            e = max_abs(out.cpu().numpy(), ref.numpy())
            print(f"\n[irsde+unet {mode}] {T}-step trajectory err {e:.3e}")
            assert e < 5e-4
    finally:
        U.FORCE_EPS = None


def test_unet_api_errors():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.unet import ConditionalUNet
    m = ConditionalUNet(3, 3, 32, [1, 2, 4, 8], 512, True, True).cuda()
    x = torch.zeros(1, 3, 32, 32, device="cuda")
    with pytest.raises(ValueError):
        m(x, x, 5, text_context=torch.zeros(1, 512, device="cuda"))       # SpatialTransformer needs image_context
    with pytest.raises(RuntimeError):
        m(x.cpu(), x.cpu(), 5)
    with pytest.raises(ValueError):
        ConditionalUNet(3, 3, 16)


def test_denoising_model_wrapper():
    """feed_data / test / get_current_visuals as U/test.py:119-127 drives them."""
    cfg, sd, model, (xt, mu, text, image), g = _setup("lin", "fp32")
    from underwater_image_restoration_amd.denoising import DenoisingModel
    from underwater_image_restoration_amd.sde import IRSDE
    sde = IRSDE(max_sigma=50, T=100, sample_T=4, schedule="cosine", eps=0.005, device="cuda")  # 4 sampling steps
    sde.set_model(model)
    dm = DenoisingModel(model)
    tape = torch.randn(4, *xt.shape, generator=torch.Generator().manual_seed(2)).cuda()
    dm.feed_data(xt, mu, GT=mu, text_context=text.cuda(), image_context=None)
    dm.test(sde, mode="posterior", noise_tape=tape)
    vis = dm.get_current_visuals()
    assert set(vis) == {"Input", "Output", "GT"} and vis["Output"].shape == xt.shape[1:] and not vis["Output"].is_cuda
    a = dm.output.clone()
    dm.test(sde, mode="anything-else-is-posterior", noise_tape=tape)
    assert torch.equal(a, dm.output)
    dm.test(sde, mode="sde", noise_tape=tape)
    assert not torch.equal(a, dm.output) and torch.isfinite(dm.output).all()
