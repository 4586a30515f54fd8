"""Golden-vector case table shared by make_golden.py (reference side) and the tests (oracle / HIP side)."""
from __future__ import annotations

import numpy as np

_BLOCK0 = "encoder_stages.0.blocks.0"
_TAPS = ["head", "stem", _BLOCK0 + ".hf_block", _BLOCK0 + ".lf_block", _BLOCK0 + ".cfc", _BLOCK0 + ".gate",
         _BLOCK0, "encoder_stages.2.blocks.0", "decoder_stages.1.blocks.0", "final_norm"]

STUDENT_CASES = {
    # full-width model (head_dim 16/32/64 like production), DA-CLIP style cond (512) + (K,7,7) masks
    "full32": dict(base_channels=64, blocks_per_level=2, cond_dim=512, num_masks=10, num_deg=10, batch=1, H=32, W=32,
                   mask_hw=(7, 7), wseed=0, xseed=1234, taps=_TAPS),
    # non-square (power-of-two sides: see oracle note on self-conjugate FFT bins), batch 2, SigLIP2-style cond (1024) + (10,224,224) masks, one block per level
    "rect": dict(base_channels=64, blocks_per_level=1, cond_dim=1024, num_masks=10, num_deg=None, batch=2, H=64, W=32,
                 mask_hw=(224, 224), wseed=1, xseed=99, taps=_TAPS),
    # V1 teacher style cond (20), no masks, narrow net (head_dim 4/8/16)
    "narrow": dict(base_channels=16, blocks_per_level=1, cond_dim=20, num_masks=0, num_deg=5, batch=2, H=64, W=64,
                   mask_hw=(7, 7), wseed=2, xseed=5, taps=_TAPS),
}


def compress(t, limit: int = 4096) -> np.ndarray:
    """Full tensor when small, else an evenly strided sample plus [mean, mean|.|] appended."""
    a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64).reshape(-1)
    if a.size <= limit:
        return a.astype(np.float32)
    step = -(-a.size // limit)
    return np.concatenate([a[::step], [a.mean(), np.abs(a).mean()]]).astype(np.float32)
