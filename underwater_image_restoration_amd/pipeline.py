"""Dataset-scale restoration loop with the conditioning encode of batch k+1 overlapped with the student forward of
batch k (no reference counterpart: the reference's loops -- `train_student.py` validation, `U/test.py:108-169` --
run DA-CLIP `encode_image` and the restoration network back to back on one stream).

Why it pays on MI355X: the DA-CLIP towers are ~200 launches of 800-row GEMMs, LayerNorms and 50-token attentions
that leave most of the 256 CUs idle for ~3 ms per batch, while the student's kernels fill the chip.  On a second HIP
stream those small kernels run in the CUs the student's launches leave free at their edges.

Stream rules of the library that make this safe: every kernel is enqueued on torch's
current stream; the library's shared scratch (`ops._ATTN_WS`, `ops._GEMM_WS`) is keyed by (device, stream); the
encoder's activation buffers belong to the encoder object and are only touched on the side stream; the features
handed over are fresh tensors, published with an event and `record_stream`."""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, Optional, Tuple

import torch


class ConditioningPrefetcher:
    """`launch(clip_in)` enqueues `encoder.encode_image(clip_in, control=True)` on a side stream and returns a handle;
    `take(handle)` makes the CURRENT stream wait for it and returns `(image_features, degra_features)`.

    One handle in flight at a time (the encoder reuses its activation buffers): launch k+1 after take k."""

    def __init__(self, encoder, device: Optional[torch.device] = None) -> None:
        self.encoder = encoder
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.stream = torch.cuda.Stream(device=self.device)
        self._pending = False

    def launch(self, clip_in: torch.Tensor):
        if self._pending:
            raise RuntimeError("ConditioningPrefetcher: take() the previous encode before launching the next one")
        main = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(main)            # clip_in (and everything before it) is ready for the side stream
        with torch.cuda.stream(self.stream):
            img_f, deg_f = self.encoder.encode_image(clip_in, control=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        clip_in.record_stream(self.stream)
        self._pending = True
        return img_f, deg_f, done

    def take(self, handle) -> Tuple[torch.Tensor, torch.Tensor]:
        img_f, deg_f, done = handle
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        img_f.record_stream(main)                # allocated on the side stream, consumed on this one
        deg_f.record_stream(main)
        self._pending = False
        return img_f, deg_f


def restore_stream(encoder, restore: Callable, batches: Iterable[tuple]) -> Iterator:
    """Yield `restore(x, z_d, masks)` for every `(x, clip_in, masks)` of `batches`, with `z_d` = DA-CLIP degradation
    features of `clip_in` (`U/test.py:114-117`), the encode of the next batch running under the current restore."""
    pf = ConditioningPrefetcher(encoder)
    it = iter(batches)
    cur = next(it, None)
    if cur is None:
        return
    handle = pf.launch(cur[1])
    while cur is not None:
        _, z_d = pf.take(handle)
        nxt = next(it, None)
        if nxt is not None:
            handle = pf.launch(nxt[1])
        yield restore(cur[0], z_d, cur[2])
        cur = nxt


class DualReplicaRestorer:
    """Two forwards of one student in flight: batches alternate between two activation-buffer sets (`model.replica()`,
    shared weights) on two HIP streams, the second released when the first has passed its first WFI block, which keeps
    the two a fraction of a forward apart.  The exp-bound attention launches of one batch then run next to the
    HBM- and LDS-bound kernels of the other.  Measured at batch 16, 256x256: 54.5 -> 52.5 ms per batch on its own, but only
    56.1 -> 55.5 ms next to the encode prefetch and the SFFB side stream, which already use most of the idle capacity,
    and the level-0 attention launch slows by 2 % (its blocks fill the register file: foreign workgroups only get in
    as its rounds retire).  Twice the activation memory.  An option for throughput-only loops; `bench.py` keeps it off.

    `submit(x, z_d, masks)` enqueues a forward and returns `(outputs, done_event)`; make the consuming stream wait for
    `done_event` (or call `wait(handle)`) before touching the outputs.  Outputs are bit-identical to `model(...)`."""

    def __init__(self, model, stagger_after: int = 1) -> None:
        self.replicas = [model, model.replica()]
        dev = next(model.parameters()).device
        self.device = dev
        self.streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        self.stagger_after = stagger_after
        self._turn = 0
        self._mid = None          # event: the forward submitted last has enqueued `stagger_after` blocks

    def submit(self, x: torch.Tensor, z_d: torch.Tensor, masks: Optional[torch.Tensor] = None):
        r = self._turn
        self._turn ^= 1
        model, st = self.replicas[r], self.streams[r]
        st.wait_stream(torch.cuda.current_stream(self.device))   # inputs are ready
        if self._mid is not None:
            st.wait_event(self._mid)
        mid = torch.cuda.Event()

        def on_block(count: int) -> None:
            if count == self.stagger_after:
                mid.record(st)
        model.on_block = on_block
        try:
            with torch.cuda.stream(st):
                out = model(x, z_d, masks)
                done = torch.cuda.Event()
                done.record(st)
        finally:
            model.on_block = None
        for t in (x, z_d, masks):
            if t is not None:
                t.record_stream(st)
        self._mid = mid
        return out, done

    def wait(self, handle):
        out, done = handle
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        restored, alphas, logits = out
        for t in [restored, logits] + list(alphas):
            if t is not None:
                t.record_stream(cur)
        return out
