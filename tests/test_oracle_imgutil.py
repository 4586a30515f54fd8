"""CPU: pin oracle/imgutil_ref.py to the reference's own tensor2img / calculate_psnr outputs (bit-exact bytes)."""
import os

import numpy as np

from oracle import imgutil_ref as R


def test_tensor2img_and_psnr_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "imgutil.npz"))
    assert np.array_equal(R.tensor2img(g["a"]), g["a_img"])
    assert np.array_equal(R.tensor2img(g["b"]), g["b_img"])
    assert np.array_equal(R.tensor2img(g["gray"]), g["gray_img"])
    assert np.array_equal(R.tensor2img(g["wide"], min_max=(-1, 1)), g["wide_img"])
    assert abs(R.calculate_psnr(g["a_img"], g["b_img"]) - float(g["psnr_ab"])) < 1e-12
    assert R.calculate_psnr(g["a_img"], g["a_img"]) == float("inf") == float(g["psnr_aa"])
    # round-half-to-even on the ties planted in the fixture: (k + 0.5) -> even neighbour (up to fp32 representability)
    row = g["a_img"][0, :8, 2]   # BGR: channel 0 of the tensor is the last byte
    assert row.dtype == np.uint8 and set(np.diff(row.astype(int))) <= {0, 1, 2}
