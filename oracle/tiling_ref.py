"""CPU oracle for halo tiling (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference has no tiling; the product defines it (underwater_image_restoration_amd/tiling.py docstring) as "the
reference forward on each window, stitched by ownership".  This restatement is deliberately written per OUTPUT
PIXEL BLOCK with its own index arithmetic (no shared helper with the product): for each core cell it derives the
window origin, runs ``forward_fn`` on that window and copies the owned region.
``forward_fn(x_window, cond, masks_window) -> restored_window`` is the oracle student (oracle/student_ref.py).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F


def tiled_forward(forward_fn: Callable, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor], tile: int,
                  halo: int) -> torch.Tensor:
    B, C, H, W = x.shape
    if H < tile or W < tile:
        return forward_fn(x, cond, masks)
    fm = None if masks is None else F.interpolate(masks, size=(H, W), mode="bilinear", align_corners=False)
    core = tile - 2 * halo
    out = torch.full_like(x, float("nan"))
    ny, nx = -(-H // core), -(-W // core)
    for iy in range(ny):
        for ix in range(nx):
            y0, x0 = iy * core, ix * core
            y1, x1 = min(y0 + core, H), min(x0 + core, W)
            wy = y0 - halo
            wy = 0 if wy < 0 else (H - tile if wy + tile > H else wy)
            wx = x0 - halo
            wx = 0 if wx < 0 else (W - tile if wx + tile > W else wx)
            xw = x[:, :, wy:wy + tile, wx:wx + tile]
            mw = None if fm is None else fm[:, :, wy:wy + tile, wx:wx + tile]
            ow = forward_fn(xw, cond, mw)
            out[:, :, y0:y1, x0:x1] = ow[:, :, y0 - wy:y1 - wy, x0 - wx:x1 - wx]
    assert not torch.isnan(out).any()
    return out
