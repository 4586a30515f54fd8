"""CPU: the oracle's restatement of the cv2-free RL enhancement operators against the
outputs of the reference's own functions (tests/golden/enhance.npz, produced by tests/golden/make_golden.py enhance)."""
import os

import numpy as np
import pytest
import torch

from oracle import enhance_ref as ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "enhance.npz"))


def t(name, dtype=torch.float32):
    return torch.from_numpy(G[name]).to(dtype)


def close(a, b, tol):
    a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
    assert a.shape == b.shape
    assert float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) <= tol, float(np.abs(a - b).max())


def test_pointwise_ops_bit_exact():
    img, other = t("img"), t("other")
    for key, gamma in (("gamma_0p7", 0.7), ("gamma_2p2", 2.2), ("gamma_floor", -1.0)):
        assert np.array_equal(ref.gamma_correction(img, gamma).numpy(), G[key])
    assert np.array_equal(ref.lf_hf_mix(img, other, 0.3).numpy(), G["mix_0p3"])
    assert np.array_equal(ref.lf_hf_mix(img, other, 1.0).numpy(), G["mix_1p0"])


def test_wavelet_denoise():
    img = t("img")
    close(ref.wavelet_denoise(img), G["wd_1p0"], 2e-6)          # fp32: conv2d vs explicit Haar sums round differently
    close(ref.wavelet_denoise(img, 2.5), G["wd_2p5"], 2e-6)
    close(ref.wavelet_denoise(img.double()), G["wd_f64"], 1e-12)
    assert float(np.abs(G["wd_2p5"] - G["img"]).max()) > 1e-2  # the threshold does something on this input


def test_guided_filter_single_channel_and_rgb_error():
    p, i = t("gf_p"), t("gf_i")
    close(ref.guided_filter(p, i), G["gf_r4"], 5e-6)
    close(ref.guided_filter(p, i, radius=2, eps=1e-2), G["gf_r2"], 5e-6)
    close(ref.guided_filter(p.double(), p.double()), G["gf_self_f64"], 1e-12)
    assert bool(G["gf_rgb_raises"])                              # the reference cannot filter an RGB image ...
    with pytest.raises(RuntimeError):
        ref.guided_filter(t("img"), t("img"))                    # ... and neither does the restatement


def test_multi_scale_retinex_separable_equals_2d():
    close(ref.multi_scale_retinex(t("small")), G["msr_small_default"], 2e-5)
    close(ref.multi_scale_retinex(t("img"), sigmas=(1.5, 4.0)), G["msr_img_s"], 2e-5)
    close(ref.multi_scale_retinex(t("img", torch.float64), sigmas=(2.0, 9.0, 15.0)), G["msr_img_f64"], 1e-9)
    assert 0.05 < float(G["msr_img_s"].std())                    # not saturated: the comparison means something


def test_state_features():
    img = t("img")
    u8 = ref.to_uint8(img)
    assert np.array_equal(u8, G["img_u8"])
    assert abs(ref.compute_uicm(u8) - float(G["uicm"])) <= 1e-4 * abs(float(G["uicm"]))   # reference sums in fp32
    assert abs(ref.compute_sharpness(img[None]) - float(G["sharpness"])) <= 1e-6
    assert abs(ref.compute_sharpness(img[None].double()) - float(G["sharpness_f64"])) <= 1e-12
    m, s = ref.histogram_stats(img[None])
    assert abs(m - float(G["hist_mean"])) <= 1e-7 and abs(s - float(G["hist_std"])) <= 1e-7
