"""CPU oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py) for
/root/reference/legacy/third_party/universal-image-restoration/utils/img_utils.py:136-163 (tensor2img, 3-D / 2-D
inputs) and :182-189 (calculate_psnr), in numpy.  Pinned by tests/golden/imgutil.npz (the reference's own functions,
tests/golden/make_golden.py imgutil)."""
from __future__ import annotations

import math

import numpy as np


def tensor2img(x: np.ndarray, min_max=(0, 1)) -> np.ndarray:
    x = np.squeeze(np.asarray(x, dtype=np.float32))
    lo, hi = np.float32(min_max[0]), np.float32(min_max[1])
    x = (np.clip(x, lo, hi) - lo) / (hi - lo)                 # fp32 throughout, like the torch ops  (:142-143)
    if x.ndim == 3:
        x = np.transpose(x[::-1], (1, 2, 0))                    # channels [2, 1, 0], CHW -> HWC       (:150-151)
    elif x.ndim != 2:
        raise TypeError("3-D or 2-D input expected")
    return np.round(x * np.float32(255.0)).astype(np.uint8)     # round half to even                    (:160-163)


def calculate_psnr(a: np.ndarray, b: np.ndarray) -> float:
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
