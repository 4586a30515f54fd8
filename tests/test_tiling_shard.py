"""CPU: tile planning properties, the tiling oracle's stitch rule, and the world_size-2 sharding paths over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import tiling_ref as TR
from underwater_image_restoration_amd import shard
from underwater_image_restoration_amd.tiling import TiledRestorer, plan_tiles


@pytest.mark.parametrize("geom", [(1080, 1920, 256, 32), (256, 256, 256, 32), (48, 72, 32, 8), (264, 520, 256, 16),
                                  (40, 40, 32, 0)])
def test_plan_covers_every_pixel_once_and_windows_fit(geom):
    H, W, tile, halo = geom
    tiles = plan_tiles(H, W, tile, halo)
    own = np.zeros((H, W), dtype=np.int32)
    for t in tiles:
        own[t.cy0:t.cy1, t.cx0:t.cx1] += 1
        assert 0 <= t.wy <= H - tile and 0 <= t.wx <= W - tile                  # window inside the frame
        assert t.wy <= t.cy0 and t.cy1 <= t.wy + tile and t.wx <= t.cx0 and t.cx1 <= t.wx + tile   # owns only what it sees
        # full halo wherever the frame allows it
        assert t.cy0 - t.wy >= min(halo, t.cy0) and t.cx0 - t.wx >= min(halo, t.cx0)
        assert (t.wy + tile) - t.cy1 >= min(halo, H - t.cy1) and (t.wx + tile) - t.cx1 >= min(halo, W - t.cx1)
    assert (own == 1).all()
    if (H, W, tile, halo) == (1080, 1920, 256, 32):
        assert len(tiles) == 60                                                  # 10 x 6 windows


def test_plan_rejects_bad_geometry():
    with pytest.raises(ValueError):
        plan_tiles(100, 100, 30, 4)      # tile % 8
    with pytest.raises(ValueError):
        plan_tiles(100, 100, 32, 16)     # no core left
    with pytest.raises(ValueError):
        plan_tiles(16, 100, 32, 4)       # frame smaller than a tile


class _CpuRestorer(TiledRestorer):
    """Test stand-in: the product's tiling / stitching logic around a window-local CPU function (so that halo
    handling is observable): out = x + mean(window) + cond[:, 0] + mean(mask window)."""

    def __init__(self, tile, halo, max_batch=3):
        super().__init__(self._fn, tile, halo, max_batch)

    @staticmethod
    def _fn(x, cond, masks):
        o = x + x.mean(dim=(1, 2, 3), keepdim=True) + cond[:, :1, None, None]
        if masks is not None:
            o = o + masks.mean(dim=(1, 2, 3), keepdim=True)
        return o, [], None

    def frame_masks(self, masks, H, W):
        return None if masks is None else F.interpolate(masks, size=(H, W), mode="bilinear", align_corners=False)


def _inputs(B=2, H=48, W=72):
    g = torch.Generator().manual_seed(5)
    return torch.rand(B, 3, H, W, generator=g), torch.randn(B, 4, generator=g), torch.rand(B, 2, 7, 7, generator=g)


@pytest.mark.parametrize("with_masks", [False, True])
def test_product_stitching_equals_oracle_rule(with_masks):
    x, cond, masks = _inputs()
    masks = masks if with_masks else None
    r = _CpuRestorer(32, 8)
    got = r(x, cond, masks)
    ref = TR.tiled_forward(lambda a, c, m: _CpuRestorer._fn(a, c, m)[0], x, cond, masks, 32, 8)
    assert torch.allclose(got, ref, atol=1e-6)
    small = x[:, :, :24, :24]
    assert torch.equal(r(small, cond, None), _CpuRestorer._fn(small, cond, None)[0])   # smaller than a tile: whole frame


def test_shard_range_partitions():
    for n in (0, 1, 5, 8, 4096):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= -(-n // w) for lo, hi in spans)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        # (1) dataset sharding, ragged: 5 images over 2 ranks, gather to rank 0, metrics all-reduce
        g = torch.Generator().manual_seed(3)
        data = torch.rand(5, 3, 8, 8, generator=g)
        target = torch.rand(5, 3, 8, 8, generator=g)
        fn = lambda lo, hi: data[lo:hi] * 0.5 + 0.25  # noqa: E731
        local, full = shard.restore_dataset_sharded(fn, 5, gather=True)
        ps, cnt = shard.psnr_sum(local, target[slice(*shard.shard_range(5))])
        tot = shard.allreduce_metrics([ps, cnt], "cpu")
        ref_ps, ref_cnt = shard.psnr_sum(data * 0.5 + 0.25, target)
        ok = abs(tot[0] - ref_ps) < 1e-9 and tot[1] == ref_cnt == 5
        if rank == 0:
            ok = ok and torch.equal(full, data * 0.5 + 0.25)
        else:
            ok = ok and full is None
        # (2) one frame, windows sharded over the ranks, stitched on rank 0
        x, cond, masks = _inputs(B=1)
        r = _CpuRestorer(32, 8)
        out = shard.restore_frame_sharded(r, x, cond, masks)
        if rank == 0:
            ok = ok and torch.allclose(out, r(x, cond, masks), atol=1e-6)
        else:
            ok = ok and out is None
        # (3) more ranks than units: empty shards must not dead-lock
        local, full = shard.restore_dataset_sharded(lambda lo, hi: data[lo:hi], 1, gather=True)
        ok = ok and local.shape[0] == (1 if rank == 0 else 0) and (rank != 0 or torch.equal(full, data[:1]))
        ret[rank] = bool(ok)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}
