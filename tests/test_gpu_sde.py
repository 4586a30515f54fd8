"""GPU parity of the IRSDE drop-in (fused HIP reverse steps through the C ABI) against the CPU oracle and the
trajectories recorded from the reference's own sde_utils.py."""
import os

import numpy as np
import pytest
import torch

from oracle import sde_ref as S
from tests.helpers import GOLDEN, max_abs

pytestmark = pytest.mark.gpu
TOL = 1e-5  # fp32 elementwise chain; values are O(1)


def _model(x, mu, t, **kw):
    return 0.3 * (x - mu) + 0.01 * torch.sin(x * 7 + float(t))


@pytest.fixture(scope="module")
def sde_mod():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd import _lib, sde
    _lib.load()
    return sde


@pytest.mark.parametrize("mode", ["sde", "posterior", "ode"])
def test_short_trajectories_match_reference(sde_mod, mode):
    g = np.load(os.path.join(GOLDEN, "sde.npz"))
    sde = sde_mod.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device="cuda")
    mu, xt = torch.from_numpy(g["traj_mu"]).cuda(), torch.from_numpy(g["traj_xt"]).cuda()
    sde.set_mu(mu)
    sde.set_model(_model)
    tape = torch.from_numpy(g[f"traj_{mode}_tape"]).cuda() if f"traj_{mode}_tape" in g.files else None
    fn = {"sde": sde.reverse_sde, "posterior": sde.reverse_posterior, "ode": sde.reverse_ode}[mode]
    out = fn(xt, T=6, noise_tape=tape) if tape is not None else fn(xt, T=6)
    assert max_abs(out.cpu().numpy(), g[f"traj_{mode}_out"]) < TOL


@pytest.mark.parametrize("mode", ["sde", "posterior"])
@pytest.mark.parametrize("sched", ["cosine", "linear"])
def test_full_100_step_loop_matches_oracle(sde_mod, mode, sched):
    """All T=100 steps at the BASELINE config-4 state shape (B=8, 3x256x256 would be 6 MB/step of tape: use 64x64)."""
    gen = torch.Generator().manual_seed(99)
    mu = torch.rand(8, 3, 64, 64, generator=gen)
    tb = S.SdeTables(50, T=100, schedule=sched, eps=0.005)
    xt = mu + torch.randn(mu.shape, generator=gen) * tb.max_sigma
    tape = torch.randn(100, *mu.shape, generator=gen)
    ref = tb.sample(_model, xt.double(), mu.double(), mode, tape.double())
    sde = sde_mod.IRSDE(max_sigma=50, T=100, schedule=sched, eps=0.005, device="cuda")
    sde.set_mu(mu.cuda())
    sde.set_model(_model)
    fn = sde.reverse_sde if mode == "sde" else sde.reverse_posterior
    out = fn(xt.cuda(), noise_tape=tape.cuda())
    # the stand-in score model is expansive (|x| grows to O(50)): fp32 chain error is judged relative to the range
    assert max_abs(out.cpu().numpy(), ref.numpy()) < 1e-5 * max(1.0, float(ref.abs().max()))
    # the caller's state is not modified; a second run is bit-identical
    out2 = fn(xt.cuda(), noise_tape=tape.cuda())
    assert torch.equal(out, out2)


def test_single_steps_reference_signatures(sde_mod):
    gen = torch.Generator().manual_seed(3)
    tb = S.SdeTables(50, T=100, schedule="cosine", eps=0.005)
    x, mu, nz, eps = (torch.randn(2, 3, 16, 16, generator=gen) for _ in range(4))
    sde = sde_mod.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device="cuda")
    sde.set_mu(mu.cuda())
    for t in (1, 37, 100):
        score = -nz / tb.sigma_bars[t]
        ref = tb.reverse_sde_step(x.double(), mu.double(), nz.double(), t, eps.double())
        got = sde.reverse_sde_step(x.cuda(), score.cuda(), t, eps=eps.cuda())
        assert max_abs(got.cpu().numpy(), ref.numpy()) < TOL
        refm = tb.reverse_sde_step(x.double(), mu.double(), nz.double(), t, None)
        assert max_abs(sde.reverse_sde_step_mean(x.cuda(), score.cuda(), t).cpu().numpy(), refm.numpy()) < TOL
        refp = tb.reverse_posterior_step(x.double(), mu.double(), nz.double(), t, eps.double())
        gotp = sde.reverse_posterior_step(x.cuda(), nz.cuda(), t, eps=eps.cuda())
        assert max_abs(gotp.cpu().numpy(), refp.numpy()) < TOL
        x0 = sde.get_init_state_from_noise(x.cuda(), nz.cuda(), t)
        A = torch.exp(tb.thetas_cumsum[t] * tb.dt)
        assert max_abs(x0.cpu().numpy(), ((x - mu - tb.sigma_bars[t] * nz) * A + mu).numpy()) < 1e-4
    with pytest.raises(RuntimeError):
        sde.reverse_posterior_step(x, nz, 5)  # CPU tensors: no fallback


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_conditioning_schedule_is_bit_identical(sde_mod, precision):
    """The sampler asks a ConditionalUNet of this package for the (time, context)-only part of all its steps in one
    batched pass (unet.conditioning_schedule).  Same rows, same kernels: the loop's result must not change by a bit, and
    calls that do not match the schedule (other time, other context tensor, tensor time) must still work."""
    from underwater_image_restoration_amd.synth import synth_state_dict, unet_param_shapes
    from underwater_image_restoration_amd.unet import ConditionalUNet
    unet = ConditionalUNet(3, 3, 32, [1, 2], 64, True, True, precision=precision)
    unet.load_state_dict(synth_state_dict(unet_param_shapes(3, 3, 32, [1, 2], 64, True, True), seed=5))
    unet = unet.cuda().eval()
    gen = torch.Generator().manual_seed(3)
    B = 3
    mu = torch.rand(B, 3, 32, 32, generator=gen).cuda()
    tc, ic = torch.randn(B, 64, generator=gen).cuda(), torch.randn(B, 64, generator=gen).cuda()
    sde = sde_mod.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device="cuda")
    sde.set_mu(mu)
    xt = mu + torch.randn(mu.shape, generator=gen).cuda() * 0.3
    tape = torch.randn(5, *mu.shape, generator=gen).cuda()
    sde.set_model(unet)
    a = sde.reverse_posterior(xt, T=5, noise_tape=tape, text_context=tc, image_context=ic)
    sde.set_model(lambda x, m, t, **kw: unet(x, m, t, **kw))       # a plain callable: no schedule, step by step
    b = sde.reverse_posterior(xt, T=5, noise_tape=tape, text_context=tc, image_context=ic)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    with unet.conditioning_schedule([1.0, 2.0], B, text_context=tc, image_context=ic):
        in_sched = unet(xt, mu, 2.0, text_context=tc, image_context=ic).clone()
        other_t = unet(xt, mu, 3.0, text_context=tc, image_context=ic).clone()          # not in the schedule
        other_c = unet(xt, mu, 2.0, text_context=tc.clone(), image_context=ic).clone()  # another tensor object
        tens_t = unet(xt, mu, torch.full((B,), 2.0), text_context=tc, image_context=ic).clone()
    assert torch.equal(in_sched, unet(xt, mu, 2.0, text_context=tc, image_context=ic))
    assert torch.equal(other_c, in_sched) and torch.equal(tens_t, in_sched)
    assert torch.equal(other_t, unet(xt, mu, 3.0, text_context=tc, image_context=ic))


def test_save_states_writes_the_reference_files(sde_mod, tmp_path):
    """sde_utils.py:269-275: every T // 100 steps the two channel halves of the state, side by side, as state_{idx}.png."""
    from PIL import Image
    import numpy as np
    gen = torch.Generator().manual_seed(8)
    mu = torch.rand(2, 6, 8, 12, generator=gen).cuda()
    sde = sde_mod.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device="cuda")
    sde.set_mu(mu)
    sde.set_model(lambda x, m, t, **kw: torch.zeros_like(x))
    d = str(tmp_path / "states")
    out = sde.reverse_ode(mu + 0.1, save_states=True, save_dir=d)
    files = sorted(os.listdir(d))
    assert len(files) == 100 and "state_1.png" in files and "state_100.png" in files
    last = np.asarray(Image.open(os.path.join(d, "state_1.png")))          # written after the final step (t = 1)
    exp = sde_mod.state_grid_uint8(torch.cat(out.chunk(2, dim=1), dim=3)).numpy()
    assert last.shape == (8 + 4, 2 * (24 + 2) + 2, 3) and np.array_equal(last, exp)
    with pytest.raises(RuntimeError):                                        # 3 channels do not split in two halves
        sde.set_mu(mu[:, :3].contiguous())
        sde.reverse_ode(mu[:, :3].contiguous(), save_states=True, save_dir=d)
