"""One-process-per-GPU sharding of the restoration path over RCCL/xGMI (``torch.distributed``; backend "nccl" on the
GPU box IS RCCL, "gloo" in the CPU tests).

The path shards by independent units -- images of a dataset (BASELINE config 5) or halo windows of one large frame
(config 3) -- so there is NO data-path collective: weights are replicated, every rank restores its contiguous
block of units, and RCCL is only used afterwards to (optionally) gather outputs to rank 0 and to all-reduce a small
fp64 metrics vector (the counterpart of the reference's only inference-side collectives,
underwater_ir/student/train_student.py:646 all_reduce and :715 barrier).  xGMI is point-to-point: the 7-to-1
gather uses each sender's own direct link to the root.
"""
from __future__ import annotations

import math
import os
import socket
import subprocess
import sys
import threading
import time
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``ceil(n/world)`` items for ``rank`` (trailing ranks may get fewer / none)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    per = math.ceil(n_items / world_size) if n_items > 0 else 0
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def gather_rows(local: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank blocks (split along dim 0 by ``shard_range``) onto ``dst``; other ranks get None.
    Blocks are padded to the common block size so a plain gather works for ragged tails."""
    rank, ws = world()
    if ws == 1:
        return local
    if local.is_cuda and dist.get_backend() == "gloo":      # rehearsals: gloo moves host memory
        out = gather_rows(local.cpu(), n_total, dst)
        return None if out is None else out.to(local.device)
    per = math.ceil(n_total / ws)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(ws):
        lo, hi = shard_range(n_total, r, ws)
        parts.append(bufs[r][:hi - lo])
    return torch.cat(parts, dim=0)


def allreduce_metrics(values: Sequence[float], device) -> List[float]:
    """SUM all-reduce of a small fp64 vector, e.g. [psnr_sum, ssim_sum, count] (train_student.py:646 uses AVG of one
    scalar; sums + count give exact dataset means for ragged shards)."""
    if world()[1] > 1 and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if world()[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def psnr_sum(restored: torch.Tensor, target: torch.Tensor) -> Tuple[float, int]:
    """Per-image PSNR 10*log10(1/mse) capped at 100 dB (train_student.py:91-95), summed over the local batch."""
    mse = ((restored.double() - target.double()) ** 2).flatten(1).mean(dim=1)
    p = torch.where(mse < 1e-10, torch.full_like(mse, 100.0), 10.0 * torch.log10(1.0 / mse.clamp_min(1e-300)))
    return float(p.sum()), int(p.numel())


@torch.no_grad()
def restore_dataset_sharded(restore_fn: Callable[[int, int], torch.Tensor], n_images: int, gather: bool = False,
                            dst: int = 0):
    """``restore_fn(lo, hi)`` restores images [lo, hi) (any batching inside) and returns them stacked on dim 0.
    Every rank handles ``shard_range(n_images)``; outputs stay sharded unless ``gather``.  -> (local, gathered)."""
    lo, hi = shard_range(n_images)
    local = restore_fn(lo, hi)
    if local.shape[0] != hi - lo:
        raise ValueError("restore_fn returned a wrong number of images")
    return local, (gather_rows(local, n_images, dst) if gather else None)


@torch.no_grad()
def restore_frame_sharded(restorer, x: torch.Tensor, cond: torch.Tensor, masks: Optional[torch.Tensor] = None,
                          dst: int = 0) -> Optional[torch.Tensor]:
    """One large frame, windows sharded over the ranks (``restorer`` is a tiling.TiledRestorer; every rank holds the
    full input frame, 24.9 MB at 1080p).  The restored windows are gathered to ``dst`` and stitched there."""
    rank, ws = world()
    H, W = x.shape[-2:]
    tiles = restorer.windows(x)
    if tiles is None or ws == 1:
        return restorer(x, cond, masks)
    lo, hi = shard_range(len(tiles))
    fm = restorer.frame_masks(masks, H, W)
    if getattr(restorer.model, "supports_windows", False) and x.is_cuda:
        # kernel-side tiling: every rank writes the cores ITS windows own into a zero frame; every pixel is owned by
        # exactly one window, so the SUM-reduce to `dst` over xGMI (24.9 MB at 1080p) IS the stitched frame
        local = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
        desc = restorer.window_descriptors(x.shape[0], tiles, x.device, (H, W)).view(x.shape[0], len(tiles), 8)[:, lo:hi]
        if hi > lo:
            restorer.restore_windows_inplace(x, cond, fm, desc.reshape(-1, 8).contiguous(), local)
        if dist.get_backend() == "gloo":                   # rehearsals: gloo moves host memory
            host = local.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            return host.to(x.device) if rank == dst else None
        dist.reduce(local, dst=dst, op=dist.ReduceOp.SUM)
        return local if rank == dst else None
    T = restorer.tile
    if hi > lo:
        win = restorer.run_windows(x, cond, fm, tiles[lo:hi])            # (B, n_local, C, T, T)
    else:
        win = torch.empty(x.shape[0], 0, x.shape[1], T, T, dtype=torch.float32, device=x.device)
    allw = gather_rows(win.transpose(0, 1).contiguous(), len(tiles), dst)  # windows on dim 0
    if rank != dst:
        return None
    return restorer.stitch(allw.transpose(0, 1), tiles, H, W)


def launch_local_ranks(script: str, argv: Sequence[str], n_ranks: int, extra_env: Optional[dict] = None,
                       deadline_s: Optional[float] = None, poll_s: float = 0.2) -> int:
    """Start ``n_ranks`` copies of ``python script *argv`` on THIS node, one per GPU, with the ``torchrun`` environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and wait for them: what
    ``python -m torch.distributed.run --nproc-per-node N`` does, for callers that were started as a bare
    ``python bench.py --gpus N``.  Must be called BEFORE the calling process touches the GPU (it only waits; a process
    that has initialised HIP must not exec, and holding a context on GPU 0 next to rank 0 would skew the run).

    Rank 0's stdout is relayed line by line by a reader thread; the ranks above 0 send their stdout to this process's
    stderr (their diagnostics stay visible, the result stream stays rank 0's); every rank inherits stderr.  The main
    thread polls ALL ranks: the first non-zero exit terminates the others at once (by PID: SIGTERM, SIGKILL after 5 s),
    so a rank that dies while its peers sit in rendezvous or a collective cannot hang the run.  ``deadline_s`` (default
    ``$UWIR_LAUNCH_DEADLINE_S`` or 3300) bounds the whole call: past it every child is killed and 124 is returned.
    The rendezvous port stays bound (SO_REUSEADDR) by this process until just before rank 0 is started.
    -> 0 when every rank exited 0, else the first non-zero exit code seen."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    if deadline_s is None:
        deadline_s = float(os.environ.get("UWIR_LAUNCH_DEADLINE_S", "3300"))
    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    procs: List[subprocess.Popen] = []

    def spawn(r: int) -> subprocess.Popen:
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n_ranks))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)
        env.update(extra_env or {})
        return subprocess.Popen([sys.executable, "-u", script, *argv], env=env,
                                stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0))

    def relay(stream) -> None:
        try:
            for line in stream:          # rank 0 prints the result line(s)
                sys.stdout.write(line)
                sys.stdout.flush()
        except ValueError:               # stream closed under us after a kill
            pass

    def stop_all() -> None:
        live = [q for q in procs if q is not None]
        for q in live:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + 5.0
        for q in live:
            try:
                q.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    t0 = time.monotonic()
    reader = None
    rc = 0
    try:
        # ranks above 0 first (they only connect), rank 0 (which binds MASTER_PORT) right after the port is released
        procs = [None] * n_ranks  # type: ignore[list-item]
        for r in range(n_ranks - 1, 0, -1):
            procs[r] = spawn(r)
        sock.close()
        procs[0] = spawn(0)
        reader = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
        reader.start()
        while True:
            codes = [q.poll() for q in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() - t0 > deadline_s:
                sys.stderr.write(f"launch_local_ranks: deadline of {deadline_s:.0f} s passed, killing {n_ranks} ranks\n")
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        try:
            sock.close()
        except OSError:
            pass
        stop_all()
        if reader is not None:
            reader.join(timeout=5.0)
    return rc
