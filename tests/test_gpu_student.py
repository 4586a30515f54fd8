"""GPU parity of the whole student forward (HIP path through the C ABI) against the fp64 CPU oracle and the
golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import student_ref as R
from tests.helpers import (STUDENT_CASES, compress, max_abs, oracle_tap_lookup, psnr, psnr_delta_at,
                           student_case_setup)

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4  # north_star: fp32 max abs error <= 1e-4 on [0,1] pixels


def _model(mk, sd, precision):
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    m = NAFNetWFIGate(**mk, precision=precision)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.cuda().eval()


def _report(tag, taps_gpu, taps_ref):
    rows = []
    for blk, bt in taps_ref.items():
        if not isinstance(bt, dict) or blk not in taps_gpu:
            continue
        for key in ("hf", "lf", "t_out", "f_out", "merged", "alpha", "out"):
            ref, got = bt[key], taps_gpu[blk][key]
            rows.append((blk, key, float((got.double().cpu() - ref.double()).abs().max()), float(ref.abs().max())))
    print(f"\n[{tag}] per-block max abs error (|ref|max):")
    for r in rows:
        print(f"   {r[0]:32s} {r[1]:7s} {r[2]:.3e}  ({r[3]:.3e})")


@pytest.mark.parametrize("name", list(STUDENT_CASES))
def test_student_fp32_matches_oracle_and_golden(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, mk, sd, x, z, m, gold = student_case_setup(name)
    sd64 = {k: v.double() for k, v in sd.items()}
    rt = {}
    with torch.no_grad():
        ro, ra, rl = R.student_forward(sd64, x.double(), z.double(), None if m is None else m.double(),
                                       blocks_per_level=cfg["blocks_per_level"], taps=rt)
    model = _model(mk, sd, "fp32")
    gt = {}
    out, alphas, logits = model(x.cuda(), z.cuda(), None if m is None else m.cuda(), taps=gt)
    torch.cuda.synchronize()
    _report(name + "/fp32", gt, rt)
    e_pre = max_abs(gt["pre_clamp"].cpu().numpy(), rt["pre_clamp"].numpy())
    e_out = max_abs(out.cpu().numpy(), ro.numpy())
    print(f"[{name}/fp32] pre-clamp err {e_pre:.3e}  out err {e_out:.3e}  vs golden(ref fp32) "
          f"{max_abs(out.cpu().numpy(), gold['out']):.3e}")
    assert out.shape == x.shape and out.dtype == torch.float32
    assert e_out < FP32_TOL and e_pre < FP32_TOL
    assert max_abs(out.cpu().numpy(), gold["out_f64"]) < FP32_TOL       # the reference itself, run in fp64
    assert max_abs(out.cpu().numpy(), gold["out"]) < 2 * FP32_TOL       # the reference in fp32 (carries its own noise)
    assert len(alphas) == len(ra)
    for a, r in zip(alphas, ra):
        assert tuple(a.shape) == tuple(r.shape)
        assert max_abs(a.cpu().numpy(), r.numpy()) < 1e-3
    if rl is None:
        assert logits is None
    else:
        assert max_abs(logits.cpu().numpy(), rl.numpy()) < 1e-3
    # second call reuses cached buffers and must be bit-identical (deterministic kernels)
    out2, _, _ = model(x.cuda(), z.cuda(), None if m is None else m.cuda())
    assert torch.equal(out, out2)


@pytest.mark.parametrize("name", ["full32", "rect", "evenodd"])
def test_student_bf16_psnr(name):
    """bf16 mode against the reference's own fp64 output: PSNR(build, ref) >= 60 dB at the golden sizes (measured
    79-80 dB), and north_star's "PSNR delta <= 0.05 dB" against a pseudo ground truth DERIVED FROM the reference output
    at 30 dB (helpers.psnr_delta_at: binds at PSNR(build, ref) ~ 49 dB; the former random-target delta could not fail)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, mk, sd, x, z, m, gold = student_case_setup(name)
    ref = torch.from_numpy(gold["out_f64"])
    model = _model(mk, sd, "bf16")
    out, alphas, logits = model(x.cuda(), z.cuda(), None if m is None else m.cuda())
    out = out.cpu()
    d = psnr_delta_at(out, ref)
    p = psnr(out, ref)
    print(f"\n[{name}/bf16] PSNR(build,ref)={p:.2f} dB  max abs err={float((out.double()-ref).abs().max()):.3e}  "
          f"PSNR delta at a 30 dB target={d:.4f} dB")
    assert d <= 0.05
    assert p >= 60.0


@pytest.mark.parametrize("attr", ["chain_blocks", "compose_reproj"])
@pytest.mark.parametrize("name", ["full32", "evenodd"])
def test_student_bf16_algebraic_shortcuts_against_the_plain_forms(name, attr):
    """Two shortcuts of bf16 mode that are identities in exact arithmetic, each against the forward without it:
    chain_blocks -- a block hands (merged, hf) to the next block of its stage instead of idwt2 -> HBM -> dwt2
    (uwir_gate_blend; dwt2 of idwt2 is the identity up to 1.2e-7 in the reference's fp32 taps; the plain stream rounds the
    full-resolution map to bf16 once more);  compose_reproj -- SFFB's reproj conv and the CFC's k / vf convs (sffb.py:55,
    cfc.py:26-28: nothing in between) run as one 1x1 conv with weights multiplied in fp64 at load time.
    Each must agree with the plain forward to bf16 noise and be about as close to the reference's fp64 output."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, mk, sd, x, z, m = student_case_setup(name)[:6]
    gold = student_case_setup(name)[6]
    ref = torch.from_numpy(gold["out_f64"])
    model = _model(mk, sd, "bf16")
    args = (x.cuda(), z.cuda(), None if m is None else m.cuda())
    assert getattr(model, attr)
    out_c, al_c, _ = model(*args)
    setattr(model, attr, False)
    out_u, al_u, _ = model(*args)
    out_c, out_u = out_c.cpu(), out_u.cpu()
    pc, pu, pcu = psnr(out_c, ref), psnr(out_u, ref), psnr(out_c, out_u.double())
    print(f"\n[{name}/bf16] {attr}: on {pc:.2f} dB, off {pu:.2f} dB against the reference; on vs off {pcu:.2f} dB")
    assert pcu >= 55.0 and pc >= pu - 1.0
    assert len(al_c) == len(al_u) and all(a.shape == b.shape for a, b in zip(al_c, al_u))


def test_student_api_errors():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    m = NAFNetWFIGate(cond_dim=8).cuda()
    with pytest.raises(ValueError, match="Condition vector z_d must be provided."):
        m(torch.zeros(1, 3, 16, 16, device="cuda"), None)
    with pytest.raises(ValueError, match="must be even"):     # the reference fails here too (out + residual, model.py:161)
        m(torch.zeros(1, 3, 21, 16, device="cuda"), torch.zeros(1, 8, device="cuda"))
    with pytest.raises(ValueError, match="too small"):
        m(torch.zeros(1, 3, 4, 16, device="cuda"), torch.zeros(1, 8, device="cuda"))


@pytest.mark.parametrize("hw", [(48, 80), (120, 72)])
def test_student_non_power_of_two_maps_match_oracle(hw):
    """Whole frames whose sub-band maps are not powers of two (48 x 80 -> 24 x 40, 12 x 20, 6 x 10; 120 x 72 -> 60 x 36,
    30 x 18, 15 x 9): the SFFB's FFT takes the mixed-radix (2^a 3^b 5^c) line kernel, or the direct DFT for other
    lengths, instead of the power-of-two one.  fp32 against the fp64 oracle, same 1e-4 bound.  (The reference's golden
    cases keep to power-of-two maps because torch's CPU FFT leaves +-1e-17 in the self-conjugate bins there)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import student_ref as R
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
    mk = dict(base_channels=16, num_levels=3, blocks_per_level=1, cond_dim=20, num_masks=10, num_degradation_types=5)
    sd = synth_state_dict(student_param_shapes(**mk), seed=21)
    model = NAFNetWFIGate(**mk, precision="fp32")
    model.load_state_dict(sd)
    model = model.cuda().eval()
    x, z, m = synth_student_inputs(2, hw[0], hw[1], 20, 10, mask_hw=(7, 7), seed=22)
    with torch.no_grad():
        ref = R.student_forward({k: v.double() for k, v in sd.items()}, x.double(), z.double(), m.double(), blocks_per_level=1)[0]
    out = model(x.cuda(), z.cuda(), m.cuda())[0]
    e = float((out.cpu().double() - ref).abs().max())
    print(f"\n[student {hw[0]}x{hw[1]} fp32] max abs err vs fp64 oracle {e:.3e}")
    assert e < 1e-4


@pytest.mark.parametrize("mk_extra", [dict(base_channels=128, num_levels=3), dict(base_channels=32, num_levels=5)])
def test_student_wide_heads_match_oracle(mk_extra):
    """Students wider / deeper than the reference's default reach head_dim 128 at their last level (base 128, 3 levels:
    hd 32 / 64 / 128; base 32, 5 levels: hd 8 .. 128): the attention kernels take hd <= 128 (the CFC's two-V pair as two
    passes above 64).  fp32 against the fp64 oracle, same bound; bf16 PSNR floor."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import student_ref as R
    from tests.helpers import psnr
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs
    mk = dict(blocks_per_level=1, cond_dim=20, num_masks=10, num_degradation_types=5, **mk_extra)
    sd = synth_state_dict(student_param_shapes(**mk), seed=31)
    x, z, m = synth_student_inputs(1, 64, 64, 20, 10, mask_hw=(7, 7), seed=32)
    with torch.no_grad():
        ref = R.student_forward({k: v.double() for k, v in sd.items()}, x.double(), z.double(), m.double(),
                                num_levels=mk["num_levels"], blocks_per_level=1)[0]
    for prec in ("fp32", "bf16"):
        model = NAFNetWFIGate(**mk, precision=prec)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        out = model(x.cuda(), z.cuda(), m.cuda())[0].cpu()
        e, p = float((out.double() - ref).abs().max()), psnr(out, ref)
        print(f"\n[student {mk_extra} {prec}] max abs err vs fp64 oracle {e:.3e}, PSNR {p:.1f} dB")
        assert e < 1e-4 if prec == "fp32" else p >= 50.0
    with pytest.raises(ValueError):
        NAFNetWFIGate(base_channels=256, num_levels=3)
