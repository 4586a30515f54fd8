"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch / numpy) of the cv2-free RL enhancement operators and state
features of the reference.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path never does.

Pinned by tests/golden/enhance.npz, which tests/golden/make_golden.py produced by importing the reference's own
`underwater_ir/rl/uiep_cwgr_ops.py`, `uiqm_uciqe.py` and `infer_adapt.py` (an empty module object stands in for the
absent `cv2` at the import statement only).

NOT restated, "parity unpinned" in this container (they call OpenCV, which the image lacks): `histogram_equalization`
(`cv2.split/merge`), `clahe_equalization`, `compute_uism` (`cv2.Sobel`), `compute_uiconm`, `compute_uciqe` and hence
`compute_uiqm` (`cv2.cvtColor` to 8-bit Lab).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .student_ref import haar_dwt2 as dwt2, haar_idwt2 as idwt2


def gamma_correction(image: torch.Tensor, gamma: float) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:114-116: exponent floored at 1e-3, input clamped to [0, 1]."""
    return image.clamp(0, 1) ** max(gamma, 1e-3)


def lf_hf_mix(low_freq: torch.Tensor, high_freq: torch.Tensor, alpha: float) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:119-120."""
    return (alpha * high_freq + (1 - alpha) * low_freq).clamp(0.0, 1.0)


def wavelet_denoise(image: torch.Tensor, threshold_scale: float = 1.0) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:46-59: one Haar level; per detail band (ALL channels together) a soft threshold
    t = scale * (median|c| / 0.6745)^2 / (sqrt(mean c^2) + 1e-6); torch.median = the LOWER middle element."""
    ll, lh, hl, hh = dwt2(image[None])
    bands = []
    for c in (lh, hl, hh):
        sigma = c.abs().flatten().sort().values[(c.numel() - 1) // 2] / 0.6745
        t = threshold_scale * (sigma ** 2 / (c.pow(2).mean().sqrt() + 1e-6))
        bands.append(torch.sign(c) * torch.relu(c.abs() - t))
    return idwt2(ll, *bands)[0]


def box_mean(x: torch.Tensor, radius: int) -> torch.Tensor:
    """zero-padded (2r+1)^2 box mean of a (1, 1, H, W) tensor (rl/uiep_cwgr_ops.py:62-65 kernel, conv2d padding=r)."""
    k = 2 * radius + 1
    return F.conv2d(x, torch.full((1, 1, k, k), 1.0, dtype=x.dtype) / (k * k), padding=radius)


def guided_filter(input_img: torch.Tensor, guidance: torch.Tensor, radius: int = 4, eps: float = 1e-3) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:68-86.  Defined for ONE channel: the reference convolves with a (1, 1, k, k) kernel at
    groups=1, so a 3-channel image raises RuntimeError there (and here)."""
    p = input_img[None] if input_img.dim() == 3 else input_img
    i = guidance[None] if guidance.dim() == 3 else guidance
    if p.shape[1] != 1 or i.shape[1] != 1:
        raise RuntimeError("guided_filter: the reference's box kernel has one input channel")
    m_i, m_p = box_mean(i, radius), box_mean(p, radius)
    a = (box_mean(i * p, radius) - m_i * m_p) / (box_mean(i * i, radius) - m_i ** 2 + eps)
    b = m_p - a * m_i
    return (box_mean(a, radius) * i + box_mean(b, radius))[0]


def multi_scale_retinex(image: torch.Tensor, sigmas=(15, 80, 250)) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:89-111: x = max(image, 1e-3); mean over sigmas of log x - log(G_sigma * x + 1e-6) with a
    zero-padded, 2-D normalised Gaussian of size 2 ceil(3 sigma) + 1; exp(.) - 1 clamped to [0, 1].  The 2-D kernel
    exp(-(u^2+v^2)/2s^2) / sum factorises exactly into two normalised 1-D kernels, which is how it is applied here
    (the reference's full 2-D convolution needs (6s+1)^2 taps per pixel)."""
    x = image.clamp_min(1e-3)[None]
    c = x.shape[1]
    acc = torch.zeros_like(x)
    for s in sigmas:
        size = int(2 * math.ceil(3 * s) + 1)
        u = torch.arange(size, dtype=torch.float64) - size // 2
        w = torch.exp(-u ** 2 / (2 * s ** 2))
        w = (w / w.sum()).to(x.dtype)
        blur = F.conv2d(x, w.view(1, 1, 1, size).repeat(c, 1, 1, 1), padding=(0, size // 2), groups=c)
        blur = F.conv2d(blur, w.view(1, 1, size, 1).repeat(c, 1, 1, 1), padding=(size // 2, 0), groups=c)
        acc += torch.log(x) - torch.log(blur + 1e-6)
    return (torch.exp(acc / len(sigmas)) - 1.0).clamp(0.0, 1.0)[0]


def to_uint8(image: torch.Tensor) -> np.ndarray:
    """rl/uiqm_uciqe.py:8-10: clamp, HWC, *255, truncate."""
    return (image.clamp(0, 1).permute(1, 2, 0).numpy() * 255).astype(np.uint8)


def compute_uicm(image_u8: np.ndarray) -> float:
    """rl/uiqm_uciqe.py:13-21 on the uint8 HWC image: rg = R - G, yb = (R + G) / 2 - B, population std."""
    r, g, b = (image_u8[:, :, i].astype(np.float64) for i in range(3))
    rg, yb = r - g, 0.5 * (r + g) - b
    return float((-0.0268 * rg.mean() + 0.1586 * rg.std()) - (0.0504 * yb.mean() - 0.164 * yb.std()))


def compute_sharpness(image: torch.Tensor) -> float:
    """rl/infer_adapt.py:52-59: mean Sobel gradient magnitude (zero padding) of the channel-mean of a (1, C, H, W) image."""
    gray = image.mean(dim=1, keepdim=True)
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=image.dtype).view(1, 1, 3, 3)
    gx, gy = F.conv2d(gray, kx, padding=1), F.conv2d(gray, kx.transpose(2, 3), padding=1)
    return float(torch.sqrt(gx ** 2 + gy ** 2).mean())


def histogram_stats(image: torch.Tensor):
    """rl/infer_adapt.py:65-66: mean and UNBIASED std over every element."""
    return float(image.mean()), float(image.std())


# ----------------------------------------------------------------------------------------------------------------------
# The OpenCV-backed functions (rl/uiep_cwgr_ops.py:24-44, rl/uiqm_uciqe.py:24-65).  PARITY UNPINNED: cv2 is absent from
# the build container, so neither these nor the kernels could be checked against the reference itself; they restate the
# arithmetic OpenCV documents (cvtColor's float L*a*b* formulas, clahe.cpp's CLAHE, Sobel with BORDER_REFLECT_101).
# histogram_equalization is the reference's own numpy code with cv2.split / cv2.merge written as channel indexing.
# ----------------------------------------------------------------------------------------------------------------------
def to_uint8_chw(image: torch.Tensor) -> np.ndarray:
    """rl/uiep_cwgr_ops.py:14-16 (HWC uint8 image)."""
    return (image.detach().cpu().clamp(0, 1).permute(1, 2, 0).numpy() * 255).astype(np.uint8)


def histogram_equalization(image: torch.Tensor) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:24-35."""
    np_img = to_uint8_chw(image)
    out = []
    for c in range(np_img.shape[2]):
        ch = np_img[:, :, c]
        hist, bins = np.histogram(ch.flatten(), 256, [0, 256])
        cdf = hist.cumsum()
        cdf = (cdf - cdf.min()) * 255 / (cdf.max() - cdf.min() + 1e-6)
        out.append(np.interp(ch.flatten(), bins[:-1], cdf).reshape(ch.shape).astype(np.uint8))
    merged = np.stack(out, axis=2)
    return torch.from_numpy(merged.astype(np.float32) / 255.0).permute(2, 0, 1).contiguous()


def _round_u8(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)        # cv::saturate_cast<uchar>(cvRound(v)): ties to even


def rgb_to_lab8(img_u8: np.ndarray) -> np.ndarray:
    """cvtColor(COLOR_RGB2LAB) on an 8-bit HWC image by the documented float formulas (float32 throughout)."""
    f32 = np.float32
    c = img_u8.astype(f32) / f32(255)
    lin = np.where(c <= f32(0.04045), c / f32(12.92), np.power((c + f32(0.055)) / f32(1.055), f32(2.4))).astype(f32)
    R, G, B = lin[..., 0], lin[..., 1], lin[..., 2]
    X = (f32(0.412453) * R + f32(0.357580) * G + f32(0.180423) * B) / f32(0.950456)
    Y = f32(0.212671) * R + f32(0.715160) * G + f32(0.072169) * B
    Z = (f32(0.019334) * R + f32(0.119193) * G + f32(0.950227) * B) / f32(1.088754)

    def f(t):
        return np.where(t > f32(0.008856), np.cbrt(t), f32(7.787) * t + f32(16.0 / 116.0)).astype(f32)
    fx, fy, fz = f(X), f(Y), f(Z)
    L = np.where(Y > f32(0.008856), f32(116) * fy - f32(16), f32(903.3) * Y).astype(f32)
    return np.stack([_round_u8(L * f32(255) / f32(100)), _round_u8(f32(500) * (fx - fy) + f32(128)),
                     _round_u8(f32(200) * (fy - fz) + f32(128))], axis=-1)


def lab8_to_rgb8(lab_u8: np.ndarray) -> np.ndarray:
    """cvtColor(COLOR_LAB2RGB) on an 8-bit HWC image (float32)."""
    f32 = np.float32
    L = lab_u8[..., 0].astype(f32) * f32(100) / f32(255)
    a, b = lab_u8[..., 1].astype(f32) - f32(128), lab_u8[..., 2].astype(f32) - f32(128)
    low = L <= f32(7.9996248)
    Y = np.where(low, L / f32(903.3), ((L + f32(16)) / f32(116)) ** 3).astype(f32)
    fy = np.where(low, f32(7.787) * Y + f32(16.0 / 116.0), (L + f32(16)) / f32(116)).astype(f32)
    fx, fz = a / f32(500) + fy, fy - b / f32(200)

    def finv(t):
        return np.where(t > f32(0.20689303), t * t * t, (t - f32(16.0 / 116.0)) / f32(7.787)).astype(f32)
    X, Z = finv(fx) * f32(0.950456), finv(fz) * f32(1.088754)
    R = f32(3.240479) * X - f32(1.53715) * Y - f32(0.498535) * Z
    G = f32(-0.969256) * X + f32(1.875991) * Y + f32(0.041556) * Z
    B = f32(0.055648) * X - f32(0.204043) * Y + f32(1.057311) * Z
    lin = np.clip(np.stack([R, G, B], axis=-1), 0, 1).astype(f32)
    s = np.where(lin <= f32(0.0031308), f32(12.92) * lin, f32(1.055) * np.power(lin, f32(1.0 / 2.4)) - f32(0.055)).astype(f32)
    return _round_u8(s * f32(255))


def clahe_u8(plane: np.ndarray, clip_limit: float, tiles_xy=(8, 8)) -> np.ndarray:
    """cv2.createCLAHE(clip_limit, tiles_xy).apply(plane) as modules/imgproc/src/clahe.cpp computes it (plain loops)."""
    H, W = plane.shape
    tx, ty = int(tiles_xy[0]), int(tiles_xy[1])
    th, tw = -(-H // ty), -(-W // tx)
    pad = np.pad(plane, ((0, th * ty - H), (0, tw * tx - W)), mode="reflect") if (th * ty != H or tw * tx != W) else plane
    area = th * tw
    clip = 0
    if clip_limit > 0:
        clip = max(int(np.float32(clip_limit) * np.float32(area) / np.float32(256)), 1)
    luts = np.zeros((ty, tx, 256), np.uint8)
    scale = np.float32(255) / np.float32(area)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(pad[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                excess = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch, residual = excess // 256, excess % 256
                hist = hist + batch
                if residual:
                    step = max(256 // residual, 1)
                    k = 0
                    while k < 256 and residual > 0:
                        hist[k] += 1
                        k += step
                        residual -= 1
            luts[j, i] = _round_u8(np.cumsum(hist).astype(np.float32) * scale)
    ys, xs = np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32)
    tyf, txf = ys * (np.float32(1) / np.float32(th)) - np.float32(0.5), xs * (np.float32(1) / np.float32(tw)) - np.float32(0.5)
    ty1, tx1 = np.floor(tyf).astype(np.int64), np.floor(txf).astype(np.int64)
    ya, xa = (tyf - ty1.astype(np.float32))[:, None], (txf - tx1.astype(np.float32))[None, :]
    ty2, tx2 = np.minimum(ty1 + 1, ty - 1), np.minimum(tx1 + 1, tx - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = plane.astype(np.int64)
    g = lambda a, b: luts[a[:, None], b[None, :], v].astype(np.float32)   # noqa: E731
    one = np.float32(1)
    res = (g(ty1, tx1) * (one - xa) + g(ty1, tx2) * xa) * (one - ya) + (g(ty2, tx1) * (one - xa) + g(ty2, tx2) * xa) * ya
    return _round_u8(res)


def clahe_equalization(image: torch.Tensor, clip_limit: float = 2.0, tile_grid_size=(8, 8)) -> torch.Tensor:
    """rl/uiep_cwgr_ops.py:38-44."""
    lab = rgb_to_lab8(to_uint8_chw(image))
    lab[:, :, 0] = clahe_u8(np.ascontiguousarray(lab[:, :, 0]), clip_limit, tile_grid_size)
    rgb = lab8_to_rgb8(lab)
    return torch.from_numpy(rgb.astype(np.float32) / 255.0).permute(2, 0, 1).contiguous()


def _sobel_reflect101(ch: np.ndarray):
    p = np.pad(ch, 1, mode="reflect")
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return gx, gy


def compute_uism(image_u8: np.ndarray) -> float:
    """rl/uiqm_uciqe.py:24-33 (cv2.Sobel ksize 3, default border BORDER_REFLECT_101)."""
    total = 0.0
    for i in range(3):
        gx, gy = _sobel_reflect101(image_u8[:, :, i].astype(np.float32))
        total += np.mean(np.sqrt(gx ** 2 + gy ** 2))
    return float(total / 3.0)


def compute_uiconm(image_u8: np.ndarray) -> float:
    """rl/uiqm_uciqe.py:36-40."""
    l_channel = rgb_to_lab8(image_u8)[:, :, 0].astype(np.float32)
    max_l, min_l = l_channel.max(), l_channel.min()
    return float((max_l - min_l) / (max_l + min_l + 1e-6))


def compute_uciqe(image_u8: np.ndarray) -> float:
    """rl/uiqm_uciqe.py:52-65."""
    lab = rgb_to_lab8(image_u8).astype(np.float32)
    l, a, b = lab[:, :, 0], lab[:, :, 1], lab[:, :, 2]
    chroma = np.sqrt(a ** 2 + b ** 2)
    return float(0.4680 * chroma.std() + 0.2745 * l.std() + 0.2576 * (chroma / (l + 1e-6)).mean())


def compute_uiqm(image: torch.Tensor) -> float:
    """rl/uiqm_uciqe.py:43-49."""
    u8 = to_uint8_chw(image)
    return 0.0282 * compute_uicm(u8) + 0.2953 * compute_uism(u8) + 3.5753 * compute_uiconm(u8)
