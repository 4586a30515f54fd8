"""Halo-overlapped tiling of large frames (BASELINE config 3: 1920x1080 on one GPU; tiles also shard across GPUs).

The reference has NO tiling: it resizes to ``img_size`` or feeds whole frames, which is
infeasible at 1080p (the CFC attention matrix alone would be ~4.3 TB).  Every global op of the student
(LayerNorm2d statistics, FFT, both attentions, channel attention) sees only its window, so tiling CHANGES the
function.  The semantics are therefore defined here, and the parity oracle (oracle/tiling_ref.py) is "the reference
forward applied to each identical window, stitched by the same rule":

  * the frame is cut into core cells of ``core = tile - 2*halo`` pixels; cell (i, j) owns the output pixels
    ``[i*core, (i+1)*core) x [j*core, (j+1)*core)`` (clipped to the frame);
  * its window is the ``tile x tile`` region around the cell (``halo`` pixels on every side), shifted so that it
    lies inside the frame; all windows have the same size, so they batch;
  * ``cond`` (z_d) is shared by all windows of a frame; masks are bilinearly resized (align_corners=False) to the
    frame once and cropped per window;
  * every output pixel is taken from the window of the cell that owns it (no blending).
A frame that is smaller than ``tile`` in either dimension is processed whole.
"""
from __future__ import annotations

from typing import List, NamedTuple, Optional, Tuple

import torch

from . import ops


class Tile(NamedTuple):
    wy: int   # window origin (inside the frame)
    wx: int
    cy0: int  # owned output region [cy0, cy1) x [cx0, cx1)
    cy1: int
    cx0: int
    cx1: int


def plan_tiles(H: int, W: int, tile: int = 256, halo: int = 16) -> List[Tile]:
    if tile % 8 or halo < 0 or tile - 2 * halo <= 0:
        raise ValueError("tile must be a multiple of 8 and larger than 2*halo")
    if H < tile or W < tile:
        raise ValueError("frame smaller than one tile: process it whole")
    core = tile - 2 * halo
    tiles = []
    for cy0 in range(0, H, core):
        cy1 = min(cy0 + core, H)
        wy = min(max(cy0 - halo, 0), H - tile)
        for cx0 in range(0, W, core):
            cx1 = min(cx0 + core, W)
            wx = min(max(cx0 - halo, 0), W - tile)
            tiles.append(Tile(wy, wx, cy0, cy1, cx0, cx1))
    return tiles


class TiledRestorer:
    """``restorer(x, cond, masks=None) -> restored`` for frames of any size (H, W multiples of 8) by running
    ``model`` (a NAFNetWFIGate) on batches of halo-overlapped windows.  ``tile_slice`` = (rank, world) restricts
    the call to this rank's share of the windows (multi-GPU: see shard.restore_frame_sharded)."""

    def __init__(self, model, tile: int = 256, halo: int = 16, max_batch: int = 32) -> None:
        # halo 16 by default: on 256-px windows it removes the seams of the local operators as well as 32 or 48 px do
        # (tools/halo_study.py: PSNR against the whole-frame result 38.3 dB without a halo, 39.35 / 39.33 / 39.32 dB at
        # 16 / 32 / 48 px -- what remains is the model's global context, which no halo restores) at 1.31x instead of
        # 1.78x recompute: 45 instead of 60 windows for a 1080p frame
        self.model, self.tile, self.halo, self.max_batch = model, tile, halo, max_batch

    def windows(self, x: torch.Tensor) -> Optional[List[Tile]]:
        H, W = x.shape[-2:]
        if H < self.tile or W < self.tile:
            return None
        return plan_tiles(H, W, self.tile, self.halo)

    @torch.no_grad()
    @ops.on_operand_device
    def frame_masks(self, masks: Optional[torch.Tensor], H: int, W: int) -> Optional[torch.Tensor]:
        """(B,M,hm,wm) -> (B,M,H,W): F.interpolate(bilinear, align_corners=False) on the device kernel."""
        if masks is None:
            return None
        B, M = masks.shape[:2]
        rows = torch.empty(B * H * W, M, dtype=torch.float32, device=masks.device)
        ops.bilinear_nchw_to_nhwc(masks.contiguous().float(), rows, H, W)
        return rows.view(B, H, W, M).permute(0, 3, 1, 2).contiguous()   # NCHW frames: the window kernels index planes

    @torch.no_grad()
    def run_windows(self, x, cond, fmasks, tiles: List[Tile]) -> torch.Tensor:
        """-> (B, len(tiles), C, tile, tile) restored windows."""
        B, T = x.shape[0], self.tile
        outs = []
        for b in range(B):
            frame_out = []
            for s in range(0, len(tiles), self.max_batch):
                chunk = tiles[s:s + self.max_batch]
                xb = torch.stack([x[b, :, t.wy:t.wy + T, t.wx:t.wx + T] for t in chunk]).contiguous()
                cb = cond[b:b + 1].expand(len(chunk), -1).contiguous()
                mb = None if fmasks is None else torch.stack(
                    [fmasks[b, :, t.wy:t.wy + T, t.wx:t.wx + T] for t in chunk]).contiguous()
                o, _, _ = self.model(xb, cb, mb)
                frame_out.append(o)  # `restored` is freshly allocated by every forward
            outs.append(torch.cat(frame_out, dim=0))
        return torch.stack(outs)

    @staticmethod
    def stitch(win: torch.Tensor, tiles: List[Tile], H: int, W: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, _, C = win.shape[:3]
        out = torch.empty(B, C, H, W, dtype=win.dtype, device=win.device) if out is None else out
        for i, t in enumerate(tiles):
            out[:, :, t.cy0:t.cy1, t.cx0:t.cx1] = win[:, i, :, t.cy0 - t.wy:t.cy1 - t.wy, t.cx0 - t.wx:t.cx1 - t.wx]
        return out

    def window_descriptors(self, n_frames: int, tiles: List[Tile], device, frame_hw=None) -> torch.Tensor:
        """int32 [n_frames * len(tiles), 8] device array of ``uwir_window`` records (frame-major), cached per plan.
        ``frame_hw`` = (FH, FW) of the frames the windows index: the records are checked against it on the host before
        they go to the device (ops.check_windows: the contract of include/uwir.h)."""
        key = (n_frames, tuple(tiles), str(device), frame_hw)
        cache = self.__dict__.setdefault("_desc_cache", {})
        d = cache.get(key)
        if d is None:
            rows = [[f, t.wy, t.wx, t.cy0, t.cy1, t.cx0, t.cx1, 0] for f in range(n_frames) for t in tiles]
            if frame_hw is not None:
                ops.check_windows(rows[:len(tiles)], n_frames, frame_hw[0], frame_hw[1], self.tile)
            d = torch.tensor(rows, dtype=torch.int32).to(device)
            if len(cache) > 16:
                cache.clear()
            cache[key] = d
        return d

    @torch.no_grad()
    def restore_windows_inplace(self, x, cond, fmasks, desc: torch.Tensor, out: torch.Tensor) -> None:
        """Kernel-side tiling: the windows ``desc`` are read in place from the frames ``x`` (and the frame-sized masks)
        by the stem / mask-resize kernels and their owned cores are written straight into ``out`` by the head kernel."""
        x = x.contiguous().float()
        fmasks = None if fmasks is None else fmasks.contiguous().float()
        for s in range(0, desc.shape[0], self.max_batch):
            d = desc[s:s + self.max_batch]
            cw = cond.index_select(0, d[:, 0].long())           # the frame's z_d for each of its windows
            self.model(x, cw, fmasks, windows=dict(win=d, T=self.tile, out=out))

    @torch.no_grad()
    @ops.on_operand_device
    def __call__(self, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.dim() != 4:
            raise ValueError("Input must be (B, C, H, W).")
        H, W = x.shape[-2:]
        if H % 8 or W % 8:
            raise ValueError("H and W must be multiples of 8")
        tiles = self.windows(x)
        if tiles is None:
            return self.model(x, cond, masks)[0]
        fm = self.frame_masks(masks, H, W)
        if getattr(self.model, "supports_windows", False) and x.is_cuda:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)   # every pixel is owned by exactly one window
            self.restore_windows_inplace(x, cond, None if fm is None else fm.contiguous(), self.window_descriptors(
                x.shape[0], tiles, x.device, (H, W)), out)
            return out
        return self.stitch(self.run_windows(x, cond, fm, tiles), tiles, H, W)   # generic callables: gather / stitch copies


class AnySizeRestorer:
    """``restorer(x, cond, masks=None) -> restored`` for frames whose H or W is NOT a multiple of 8 (540 x 960, 1080 x
    1918, ...): reflect-pad bottom / right to the next multiple of 8, restore (through ``TiledRestorer`` when the padded
    frame is at least one tile), crop.  Aligned sizes pass straight through, bit-identical to the wrapped call.

    A frame with even sides that fits one tile runs whole through the model, which follows the reference at any even size
    (odd deeper levels, the decoder's resize to the skip: S/model.py:150-151, golden case ``evenodd``).  Larger frames
    need windows on the 8-pixel grid, hence the padding; parity unpinned for those (the reference has no tiling, and at an
    odd H or W its own forward fails at ``out + residual``): only the <= 7 mirrored rows / columns of context differ from
    the aligned pipeline.  Masks are frame-level maps that the model resizes to the (padded) frame itself."""

    def __init__(self, model, tile: int = 256, halo: int = 16, max_batch: int = 32) -> None:
        self.tiled = TiledRestorer(model, tile, halo, max_batch)

    @torch.no_grad()
    @ops.on_operand_device
    def __call__(self, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.dim() != 4:
            raise ValueError("Input must be (B, C, H, W).")
        H, W = x.shape[-2:]
        ph, pw = (-H) % 8, (-W) % 8
        if ph == 0 and pw == 0:
            return self.tiled(x, cond, masks)
        if H % 2 == 0 and W % 2 == 0 and H <= self.tiled.tile and W <= self.tiled.tile:
            return self.tiled.model(x, cond, masks)[0]          # native: the reference's own path for such sizes
        if ph >= H or pw >= W:
            raise ValueError("image too small for reflect padding")
        xp = torch.nn.functional.pad(x.float(), (0, pw, 0, ph), mode="reflect")
        return self.tiled(xp, cond, masks)[:, :, :H, :W].contiguous()

