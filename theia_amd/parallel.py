"""Data-parallel training over RCCL/xGMI: one process per GPU, replicated parameters, image minibatch sharded by rank.

Replaces ``torch.nn.parallel.DistributedDataParallel`` at the reference's call site (scripts/train/train_rvfm.py:258)
for the hot path.  The reference's DDP reducer copies gradients into 25 MiB buckets as autograd hooks fire; here the
engine already WRITES gradients into flat per-bucket buffers in backward-completion order (translator heads first,
then ViT layer groups from the top down), so a bucket's all-reduce is issued -- on a side HIP stream, fenced by an
event -- the moment the engine finishes it, overlapping the rest of backward.  Averaging uses RCCL's AVG reduction
(no extra scale kernel).  Loss scalars are not reduced (the reference logs rank-local values).

``GradBucketReducer`` is backend-agnostic (RCCL on GPU, gloo on CPU) so the N>1 logic is unit-tested with
world_size-2 gloo processes; ``TheiaDataParallel`` wires it to a ``RobotVisionFM``.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    """Average flat gradient buckets across ranks, asynchronously, in the order they become ready."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._pending: List = []
        self._side: Optional[torch.cuda.Stream] = None

    def bucket_ready(self, flat: torch.Tensor, also_after: Optional["torch.cuda.Event"] = None) -> None:
        """flat is complete once the current stream -- and ``also_after`` (an event on another producer stream: the
        engine's weight-gradient queue) -- have been reached; the all-reduce waits for both on its own stream."""
        if self.world == 1:
            return
        # RCCL has an AVG reduction (no extra scale kernel); gloo (CPU tests, and the 2-ranks-on-one-GPU test) sums and
        # the result is scaled when the bucket is waited for
        avg = dist.get_backend(self.pg) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                if also_after is not None:
                    self._side.wait_event(also_after)
                # gloo stages device tensors through the host on its own pool streams; issued asynchronously from here it
                # dead-locked sporadically in work.wait() (2 ranks sharing one GPU), so that test-only combination runs
                # synchronously.  RCCL collectives are stream-ordered and stay asynchronous.
                work = dist.all_reduce(flat, op=op, group=self.pg, async_op=avg or os.environ.get("THEIA_GLOO_ASYNC") == "1")
            self._pending.append((work, flat, not avg))
        else:
            work = dist.all_reduce(flat, op=op, group=self.pg, async_op=True)
            self._pending.append((work, flat, not avg))

    def finish(self) -> None:
        """Make the current stream (GPU) / the caller (CPU) wait for every outstanding bucket."""
        for work, flat, scale in self._pending:
            if work is not None:
                work.wait()
            if scale:
                flat.div_(self.world)
        self._pending.clear()


def broadcast_parameters(params, src: int = 0, process_group=None) -> None:
    """Parameter broadcast from rank 0 at start-up (DDP constructor behaviour, train_rvfm.py:258)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for p in params:
            dist.broadcast(p.data, src=src, group=process_group)


class TheiaDataParallel(torch.nn.Module):
    """``DDP``-shaped wrapper: exposes ``.module``, ``parameters()``, ``train()/eval()``, ``__call__``."""

    def __init__(self, module: torch.nn.Module, process_group=None, broadcast: bool = True):
        super().__init__()
        self.module = module
        self.reducer = GradBucketReducer(process_group)
        if broadcast:
            broadcast_parameters(module.parameters(), 0, process_group)
        self._callback_queued = False
        if self.reducer.world > 1:
            module.engine.bucket_ready_hook = self._on_bucket

    def _on_bucket(self, bucket, side_event=None) -> None:
        if not self._callback_queued:
            # runs once when the current backward pass has finished (same mechanism DDP uses)
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            self._callback_queued = True
        self.reducer.bucket_ready(bucket.flat, side_event)

    def _finalize(self) -> None:
        self.reducer.finish()
        self._callback_queued = False

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
