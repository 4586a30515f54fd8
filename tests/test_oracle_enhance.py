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


# ---- the OpenCV-backed functions: PARITY UNPINNED (cv2 is absent here).  What CAN be checked on the CPU: the restatement
# against the widely published 8-bit Lab values of the sRGB primaries, the reference's numpy hist-eq arithmetic on a case
# whose answer is known in closed form, and structural properties of CLAHE.
def test_lab8_known_answers_and_round_trip():
    prim = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    lab = ref.rgb_to_lab8(prim)[0]
    # cv2.cvtColor(..., COLOR_RGB2LAB) of white / red / green / blue / black / mid grey as OpenCV's documentation and
    # countless posts quote them
    assert lab.tolist() == [[255, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20], [0, 128, 128], [137, 128, 128]]
    g = np.random.default_rng(0)
    grey = np.repeat(g.integers(0, 256, (16, 16, 1), dtype=np.uint8), 3, axis=2)
    back = ref.lab8_to_rgb8(ref.rgb_to_lab8(grey))
    assert int(np.abs(back.astype(int) - grey.astype(int)).max()) <= 1          # greys survive the 8-bit round trip


def test_histogram_equalization_closed_form():
    # every level 0..255 exactly four times: cdf = 4 (v + 1), (cdf - 4) * 255 / (1020 + 1e-6) in float64, truncated
    v = np.repeat(np.arange(256, dtype=np.float32), 4).reshape(32, 32)
    img = torch.from_numpy((v + 0.5) / 255.0)[None].repeat(3, 1, 1)
    out = ref.histogram_equalization(img)
    lev = np.rint(out.numpy() * 255).astype(int)[0]
    expect = np.floor((4.0 * v.astype(np.float64)) * 255 / (1020 + 1e-6)).astype(int)      # = max(v - 1, 0): the 1e-6 bites
    assert np.array_equal(lev, expect) and out.shape == img.shape


def test_clahe_properties():
    g = np.random.default_rng(1)
    plane = g.integers(0, 256, (40, 56), dtype=np.uint8)
    out = ref.clahe_u8(plane, 2.0, (8, 8))                      # 56 / 8 and 40 / 8 exact; no padding
    assert out.shape == plane.shape and out.dtype == np.uint8
    # a huge clip limit on one tile is plain equalisation: LUT = round(cdf * 255 / area), monotone in the level
    one = ref.clahe_u8(plane, 1e9, (1, 1))
    cdf = np.cumsum(np.bincount(plane.ravel(), minlength=256))
    assert np.array_equal(one, np.clip(np.rint(cdf.astype(np.float32) * np.float32(255) / np.float32(plane.size)), 0, 255).astype(np.uint8)[plane])
    # padded case (45 x 70 on an 8 x 8 grid) runs and keeps the shape
    assert ref.clahe_u8(g.integers(0, 256, (45, 70), dtype=np.uint8), 2.0).shape == (45, 70)
