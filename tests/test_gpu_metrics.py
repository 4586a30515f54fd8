"""GPU parity of the device metrics (one fused SSIM/MSE pass through the C ABI) against the CPU oracle and the
reference's own values."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as M
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


def test_metrics_match_reference_and_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd import metrics
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    for name in ("noisy", "random", "identical", "flat", "small"):
        x, y = torch.from_numpy(g[f"{name}_x"]), torch.from_numpy(g[f"{name}_y"])
        got = float(metrics.ssim_loss(x.cuda(), y.cuda()))
        assert abs(got - float(g[f"{name}_ssim_loss"])) < 1e-6, (name, got)
        assert abs(metrics.psnr_value(x.cuda(), y.cuda()) - float(g[f"{name}_psnr"])) < 1e-4, name
        per = metrics.batch_metrics(x.cuda(), y.cuda()).cpu()
        for b in range(x.shape[0]):   # per-image values
            assert abs(float(per[b, 0]) - float(M.ssim_loss(x[b:b + 1].double(), y[b:b + 1].double()))) < 1e-6
            assert abs(float(per[b, 2]) - M.psnr_value(x[b:b + 1].double(), y[b:b + 1].double())) < 1e-4
    # a size that is not a tile multiple, larger batch
    gen = torch.Generator().manual_seed(9)
    x, y = torch.rand(5, 3, 70, 45, generator=gen), torch.rand(5, 3, 70, 45, generator=gen)
    assert abs(float(metrics.ssim_loss(x.cuda(), y.cuda())) - float(M.ssim_loss(x.double(), y.double()))) < 1e-6
    assert metrics.ssim_value(x.cuda(), x.cuda()) > 0.999999 and metrics.psnr_value(x.cuda(), x.cuda()) == 100.0
    with pytest.raises(RuntimeError):
        metrics.ssim_loss(x, y)


def test_eval_batch_from_files(tmp_path):
    """f4 + f2 together: pseudo-label files -> model dims -> checkpoint (strict=False) -> forward -> PSNR/SSIM sums."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd import formats
    from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict
    gen = torch.Generator().manual_seed(4)
    pl = [{"z_d": torch.randn(20, generator=gen), "masks": torch.rand(10, 7, 7, generator=gen),
           "global_prob": torch.rand(10, generator=gen)} for _ in range(2)]
    for i, p in enumerate(pl):
        torch.save(p, tmp_path / f"im{i}.pt")
    loaded = [formats.load_pseudo_label(tmp_path, f"im{i}.png") for i in range(2)]
    sd = synth_state_dict(student_param_shapes(cond_dim=20, num_masks=10, num_degradation_types=10), seed=1)
    sd["extra.unused"] = torch.zeros(1)                 # strict=False tolerates foreign keys (infer_adapt.py:159)
    torch.save(sd, tmp_path / "student.pth")
    model, z1, m1 = formats.build_student_from_files(loaded[0], tmp_path / "student.pth")
    assert z1.shape == (1, 20) and m1.shape == (1, 10, 7, 7)
    lq, gt = torch.rand(2, 3, 32, 32, generator=gen).cuda(), torch.rand(2, 3, 32, 32, generator=gen).cuda()
    out, psnr_sum, ssim_sum, count = formats.eval_batch(model, lq, gt, loaded)
    ref_p = sum(M.psnr_value(out[i:i + 1].double().cpu().clamp(0, 1), gt[i:i + 1].double().cpu()) for i in range(2))
    ref_s = sum(M.ssim_value(out[i:i + 1].double().cpu().clamp(0, 1), gt[i:i + 1].double().cpu()) for i in range(2))
    assert count == 2 and abs(psnr_sum - ref_p) < 1e-3 and abs(ssim_sum - ref_s) < 1e-5


def test_tensor2img_and_uint8_psnr_bit_exact():
    """uint8 quantisation and the integer PSNR sum are byte / integer work: bit-exact against the reference's outputs."""
    from oracle import imgutil_ref as R
    from underwater_image_restoration_amd import imgutil
    g = np.load(os.path.join(GOLDEN, "imgutil.npz"))
    ia = imgutil.tensor2img(torch.from_numpy(g["a"]).cuda())
    ib = imgutil.tensor2img(torch.from_numpy(g["b"]).cuda())
    assert ia.dtype == np.uint8 and np.array_equal(ia, g["a_img"]) and np.array_equal(ib, g["b_img"])
    assert np.array_equal(imgutil.tensor2img(torch.from_numpy(g["gray"]).cuda()), g["gray_img"])
    assert np.array_equal(imgutil.tensor2img(torch.from_numpy(g["wide"]).cuda(), min_max=(-1, 1)), g["wide_img"])
    assert imgutil.calculate_psnr(ia, ib) == float(g["psnr_ab"])          # same integer sum, same float expression
    assert imgutil.calculate_psnr(ia, ia) == float("inf")
    # a full 1080p frame: property = agreement with the oracle on the same bytes
    gen = torch.Generator().manual_seed(5)
    big = torch.rand(1, 3, 1080, 1920, generator=gen) * 1.2 - 0.1
    dev = imgutil.tensor2img_device(big.cuda())
    ref = R.tensor2img(big.numpy())
    assert np.array_equal(dev.cpu().numpy(), ref)
    noisy = R.tensor2img((big + 0.01 * torch.randn(big.shape, generator=gen)).numpy())
    assert imgutil.calculate_psnr(dev, torch.from_numpy(noisy).cuda()) == R.calculate_psnr(ref, noisy)
    with pytest.raises(NotImplementedError):
        imgutil.tensor2img(torch.rand(2, 3, 8, 8).cuda())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        imgutil.tensor2img(torch.rand(3, 8, 8))
