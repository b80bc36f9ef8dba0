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


DEFAULT_RCCL_CHANNELS = 0  # opt-in: no cap / no CU reservation until an 8-GPU A/B says otherwise (THEIA_RCCL_MAX_NCHANNELS=16 to try)


def rccl_channels() -> int:
    """Channels (= workgroups, = CUs held for the length of a collective) RCCL may use: ``THEIA_RCCL_MAX_NCHANNELS`` (0 = leave RCCL's
    own default, no CU reservation), else a ``NCCL_MAX_NCHANNELS`` the user exported, else ``DEFAULT_RCCL_CHANNELS`` (0: both the
    channel cap and the CU reservation are OFF unless asked for -- neither has been measured at N > 1)."""
    n = os.environ.get("THEIA_RCCL_MAX_NCHANNELS")
    if n is not None and n != "":
        return max(0, int(n))
    n = os.environ.get("NCCL_MAX_NCHANNELS")
    return int(n) if n else DEFAULT_RCCL_CHANNELS


def configure_rccl_env() -> None:
    """Environment RCCL reads when the communicator is created -- call before ``init_process_group``.

    Opt-in (``THEIA_RCCL_MAX_NCHANNELS=16``): caps the number of channels RCCL runs a collective with (``NCCL_MAX_NCHANNELS`` =
    ``rccl_channels()``; default 0 = RCCL's own choice, nothing exported).
    The bucket all-reduces overlap the backward GEMMs; the persistent NT kernel and the weight-gradient kernel run one workgroup per
    CU with all of its registers and 130-150 KB of its LDS, so an RCCL workgroup cannot share a CU with them: every CU a collective
    holds pushes one workgroup of a 256-workgroup launch into a second round.  The two sides are therefore given disjoint CU sets:
    RCCL at most ``rccl_channels()`` CUs, the GEMM planners the rest while a gradient exchange is in flight (``TheiaDataParallel``
    -> ``theia_set_compute_cus``).  On xGMI a ring is bound per link (~153 GB/s), not by how many CUs copy, so few channels for
    longer is the cheap side of that trade.  ``THEIA_RCCL_MAX_NCHANNELS=0``: RCCL's defaults, nothing reserved."""
    n = rccl_channels()
    if n > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))


def reserved_cus() -> int:
    """CUs left to RCCL while gradient buckets are being exchanged: ``THEIA_DP_RESERVED_CUS`` or ``rccl_channels()``."""
    n = os.environ.get("THEIA_DP_RESERVED_CUS")
    return max(0, int(n)) if n else rccl_channels()


class AbiCommunicator:
    """``theia_comm_*`` of include/theia_hip.h: RCCL driven through the library's C ABI instead of ``torch.distributed`` -- the path a
    non-PyTorch host would use.  Rank 0's 128-byte communicator id travels over the process group that already exists
    for the rendezvous (its store); the collectives themselves are enqueued on the CURRENT stream, in place, and return nothing to wait
    for (stream order is the synchronisation)."""

    def __init__(self, process_group=None, device: Optional[torch.device] = None):
        import ctypes

        from . import _native as N
        self._N = N
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ident = ctypes.create_string_buffer(N.COMM_ID_BYTES)
        if self.rank == 0:
            N.check(N.lib().theia_comm_unique_id(ident), "theia_comm_unique_id")
        box = [ident.raw if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=process_group)
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().theia_comm_init(ctypes.byref(self._handle), box[0], self.world, self.rank), "theia_comm_init")

    def _args(self, t: torch.Tensor):
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)):
            raise ValueError("AbiCommunicator: contiguous f32 / bf16 device tensors only")
        return self._handle, t.data_ptr(), t.numel(), self._N.dtype_code(t.dtype)

    def allreduce(self, t: torch.Tensor, average: bool = True) -> None:
        h, p, n, dt = self._args(t)
        self._N.check(self._N.lib().theia_comm_allreduce(h, p, n, dt, int(average), self._N.stream_ptr()), "theia_comm_allreduce")

    def broadcast(self, t: torch.Tensor, root: int = 0) -> None:
        h, p, n, dt = self._args(t)
        self._N.check(self._N.lib().theia_comm_broadcast(h, p, n, dt, root, self._N.stream_ptr()), "theia_comm_broadcast")

    def close(self) -> None:
        if self._handle:
            torch.cuda.synchronize(self.device)
            self._N.check(self._N.lib().theia_comm_destroy(self._handle), "theia_comm_destroy")
            self._handle = None


class GradBucketReducer:
    """Average flat gradient buckets across ranks, asynchronously, in the order they become ready.

    exchange:   "allreduce" (default) -- one all-reduce (AVG) per bucket;
                "rs_ag" -- reduce-scatter (AVG) into this rank's 1/world shard of the bucket, then all-gather, both in place on the
                flat buffer (what a ring all-reduce does internally; as two collectives the shard is available in between -- the hook a
                sharded optimizer step would use -- and the all-gather can trail the next bucket's reduce-scatter);
    comm_dtype: "fp32" (default) or "bf16" -- the bucket is rounded to bf16 for the exchange (half the bytes on the xGMI links: 376
                instead of 752 MB per step for DeiT-base + 5 teachers) and widened back into the fp32 bucket afterwards.
    backend:    "torch" (default) -- ``torch.distributed`` collectives on the process group (RCCL on GPUs, gloo in the CPU tests);
                "abi" -- device buckets go through ``theia_comm_allreduce`` (``AbiCommunicator``; all-reduce exchange only), the
                process group is used for the rendezvous alone.
    Environment: THEIA_DP_EXCHANGE, THEIA_DP_COMM_DTYPE, THEIA_DP_BACKEND."""

    def __init__(self, process_group=None, exchange: Optional[str] = None, comm_dtype: Optional[str] = None,
                 backend: Optional[str] = None):
        self.pg = process_group
        self.backend = backend or os.environ.get("THEIA_DP_BACKEND", "torch")
        if self.backend not in ("torch", "abi"):
            raise ValueError(f"GradBucketReducer: backend={self.backend!r}")
        self._abi: Optional[AbiCommunicator] = None
        self._owns_abi = False
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.exchange = exchange or os.environ.get("THEIA_DP_EXCHANGE", "allreduce")
        self.comm_dtype = comm_dtype or os.environ.get("THEIA_DP_COMM_DTYPE", "fp32")
        if self.exchange not in ("allreduce", "rs_ag") or self.comm_dtype not in ("fp32", "bf16"):
            raise ValueError(f"GradBucketReducer: exchange={self.exchange!r} comm_dtype={self.comm_dtype!r}")
        self._pending: List = []
        self._side: Optional[torch.cuda.Stream] = None
        self._stage: dict = {}  # bf16 staging buffers, one per bucket (keyed by the flat buffer's address)
        self._pad: dict = {}    # rs_ag: padded staging buffers for buckets whose length is not a multiple of world

    def connect(self, device) -> None:
        """backend 'abi': create the RCCL communicator NOW -- a collective rendezvous (``broadcast_object_list`` of the id +
        ``ncclCommInitRank``) that every rank must reach together, so it belongs in the constructor of the wrapper, not inside the
        first bucket's hook in the middle of a backward pass (a rank whose backward raised would leave the others hanging there)."""
        if self.backend == "abi" and self._abi is None and self.world > 1 and torch.device(device).type == "cuda":
            self._abi = AbiCommunicator(self.pg, torch.device(device))
            self._owns_abi = True

    def close(self) -> None:
        """destroy the C-ABI communicator this reducer created (``theia_comm_destroy``); idempotent.  A communicator handed in from
        outside (``reducer._abi = comm``) belongs to its creator and is only dropped."""
        abi, self._abi = self._abi, None
        owned, self._owns_abi = getattr(self, "_owns_abi", False), False
        if abi is not None and owned:
            abi.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ the exchange itself (on whatever stream is current)
    def _exchange(self, buf: torch.Tensor, avg: bool):
        """all-reduce `buf` (SUM or AVG) with the configured collective(s); returns the last work handle (or None)"""
        if self.backend == "abi" and buf.is_cuda:
            if self.exchange != "allreduce":
                raise ValueError("GradBucketReducer: backend 'abi' exchanges with all-reduce only")
            if self._abi is None:  # (a reducer used without TheiaDataParallel: connect() was not called by a constructor)
                self.connect(buf.device)
            self._abi.allreduce(buf, average=avg)  # stream-ordered on the current (side) stream: nothing to wait for
            return None
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        async_ok = avg or not buf.is_cuda or os.environ.get("THEIA_GLOO_ASYNC") == "1"
        if self.exchange == "rs_ag":
            n = buf.numel()
            if n % self.world != 0:
                # a bucket whose length is not a multiple of world: exchange a zero-padded copy and copy the head back.  (It used to
                # fall back to an all-reduce silently; the shard boundaries must be the same on every rank, and zeros are neutral.)
                key = (buf.data_ptr(), n, buf.dtype)
                pad = self._pad.get(key)
                if pad is None:
                    pad = self._pad[key] = torch.zeros((n + self.world - 1) // self.world * self.world, dtype=buf.dtype, device=buf.device)
                pad[:n].copy_(buf)
                work = self._exchange(pad, avg)
                if work is not None:
                    work.wait()  # stream-level on RCCL; the copy below is ordered behind it
                buf.copy_(pad[:n])
                return None
            shards = buf.view(self.world, -1)
            mine = shards[self.rank]
            if dist.get_backend(self.pg) == "nccl":
                dist.reduce_scatter_tensor(mine, buf, op=op, group=self.pg, async_op=False)  # in place: output = input shard `rank`
                return dist.all_gather_into_tensor(buf, mine, group=self.pg, async_op=True)
            # gloo (CPU tests) has no reduce-scatter: one reduce per shard to its owner, then the all-gather
            for r in range(self.world):
                dist.reduce(shards[r], dst=r, op=op, group=self.pg)
            return dist.all_gather_into_tensor(buf, mine.clone(), group=self.pg, async_op=async_ok)
        return dist.all_reduce(buf, op=op, group=self.pg, async_op=async_ok)

    def _reduce(self, flat: torch.Tensor, avg: bool):
        """-> (work, post) ; post() finishes the bucket once work is done (widening / scaling)"""
        scale = 1.0 if avg else 1.0 / self.world
        if self.comm_dtype == "bf16":
            st = self._stage.get(flat.data_ptr())
            if st is None or st.numel() != flat.numel():
                st = self._stage[flat.data_ptr()] = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            if flat.is_cuda:
                from . import ops
                ops.cast(flat, st)
            else:
                st.copy_(flat)
            work = self._exchange(st, avg)

            def post():
                if flat.is_cuda:
                    from . import ops
                    ops.upcast_scale(st, flat, scale)
                else:
                    flat.copy_(st.float().mul_(scale))
            return work, post
        work = self._exchange(flat, avg)
        return work, (None if avg else (lambda: flat.div_(self.world)))

    def bucket_ready(self, flat: torch.Tensor, also_after: Optional["torch.cuda.Event"] = None) -> None:
        """flat is complete once the current stream -- and ``also_after`` (an event on another producer stream: the
        engine's weight-gradient queue) -- have been reached; the exchange waits for both on its own stream."""
        if self.world == 1:
            return
        # RCCL has an AVG reduction (no extra scale kernel); gloo (CPU tests, and the 2-ranks-on-one-GPU test) sums and
        # the result is scaled when the bucket is waited for
        avg = dist.get_backend(self.pg) == "nccl" or (self.backend == "abi" and flat.is_cuda)
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
                work, post = self._reduce(flat, avg)
                if avg and post is not None:  # RCCL + bf16 exchange: widen on the side stream, behind the collective
                    if work is not None:
                        work.wait()  # (stream-level: the side stream waits for RCCL's stream; the host does not block)
                    post()
                    work = post = None
            self._pending.append((work, flat, post))
        else:
            work, post = self._reduce(flat, avg)
            self._pending.append((work, flat, post))

    def finish(self) -> None:
        """Make the current stream (GPU) / the caller (CPU) wait for every outstanding bucket."""
        for work, flat, post in self._pending:
            if work is not None:
                work.wait()
            if post is not None:
                post()
        if self._side is not None and self._pending:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
        self._pending.clear()


def broadcast_parameters(params, src: int = 0, process_group=None, bucket_bytes: int = 256 << 20) -> None:
    """Parameter broadcast from rank 0 at start-up (DDP constructor behaviour, train_rvfm.py:258) -- coalesced: parameters are packed
    into flat buffers of up to ``bucket_bytes`` per dtype and each buffer is ONE collective (DDP coalesces the same way; one
    broadcast per tensor is ~230 latency-bound collectives for DeiT-base + 5 heads)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        group: List[torch.Tensor] = []
        size = 0

        def flush():
            nonlocal group, size
            if not group:
                return
            flat = torch.cat([p.data.reshape(-1) for p in group])
            dist.broadcast(flat, src=src, group=process_group)
            off = 0
            for p in group:
                p.data.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            group, size = [], 0

        for p in params:
            if group and (p.dtype != group[0].dtype or p.device != group[0].device or size + p.numel() * p.element_size() > bucket_bytes):
                flush()
            group.append(p)
            size += p.numel() * p.element_size()
        flush()


def pick_reservation(ms_per_step: dict, min_gain: float = 0.01) -> int:
    """The CU reservation to keep after ``TheiaDataParallel.autotune_reserved_cus``: the fastest measured one, but the smallest candidate
    (normally 0 = nothing reserved) unless another one beats it by more than ``min_gain`` (relative) -- run-to-run noise must not
    switch a reservation on."""
    base = min(ms_per_step)
    best = min(ms_per_step, key=lambda r: (ms_per_step[r], r))
    return best if ms_per_step[best] < ms_per_step[base] * (1.0 - min_gain) else base


class TheiaDataParallel(torch.nn.Module):
    """``DDP``-shaped wrapper: exposes ``.module``, ``parameters()``, ``train()/eval()``, ``__call__``.

    While gradient buckets are being exchanged (from the first completed bucket of a backward pass to the end of that pass) the GEMM
    planners are told to leave ``reserved_cus()`` CUs to RCCL (``theia_set_compute_cus``); forward passes use the whole device."""

    def __init__(self, module: torch.nn.Module, process_group=None, broadcast: bool = True):
        super().__init__()
        self.module = module
        self.reducer = GradBucketReducer(process_group)
        if broadcast:
            broadcast_parameters(module.parameters(), 0, process_group)
        self._callback_queued = False
        self._reserve = 0
        self.dynamic_schedule = False            # set by autotune_reserved_cus when the work-conserving tile schedule wins on this job
        self.autotune_dynamic_ms: Optional[float] = None
        self._saved_cus: Optional[int] = None  # the budget in force before this wrapper shrank it (None: not shrunk)
        self._saved_sched: Optional[bool] = None  # the tile schedule in force before a backward window switched the dynamic one on
        if self.reducer.world > 1:
            module.engine.bucket_ready_hook = self._on_bucket
            first = next(iter(module.parameters()), None)
            if first is not None and first.is_cuda:
                self.reducer.connect(first.device)  # collective: every rank constructs its wrapper at the same point
            if first is not None and first.is_cuda and dist.get_backend(process_group) == "nccl":
                self._reserve = reserved_cus()

    def _shrink_cus(self) -> None:
        """leave ``self._reserve`` CUs of the budget currently in force (the device, or a THEIA_COMPUTE_CUS the user set) to RCCL"""
        from . import ops
        if self._saved_cus is None:
            self._saved_cus = ops.get_compute_cus()
            ops.set_compute_cus(max(64, self._saved_cus - self._reserve))

    def _restore_cus(self) -> None:
        """put back exactly the budget that was in force before ``_shrink_cus`` (not "whole device")"""
        from . import ops
        if self._saved_cus is not None:
            ops.set_compute_cus(self._saved_cus)
            self._saved_cus = None

    def autotune_reserved_cus(self, step_fn: Callable[[], object], candidates=(0, 16, 32, 64), steps: int = 5,
                              min_gain: float = 0.01, try_dynamic: bool = True) -> dict:
        """Measure, at start-up, how many CUs the GEMM planners should leave to RCCL while gradient buckets are in flight, and keep the
        best: ``step_fn()`` (one whole training step) is timed ``steps`` times per candidate (max over ranks), ``pick_reservation``
        decides.  The persistent NT kernel and the weight-gradient kernel hold a whole CU per workgroup, so a collective that takes k CUs
        pushes k workgroups of a 256-workgroup launch into a second round unless the planners were told (``theia_set_compute_cus``); how
        many CUs RCCL actually takes depends on its version, the topology and the message size -- so it is measured on the job's own
        step instead of guessed.  Collective: every rank calls it at the same point; all ranks reach the same decision (the timings
        are reduced with MAX).  Returns {candidate: ms per step}; {} when there is nothing to tune (one rank, CPU / gloo, or the user
        fixed the value with THEIA_DP_RESERVED_CUS / THEIA_RCCL_MAX_NCHANNELS)."""
        import time
        first = next(iter(self.module.parameters()), None)
        if (self.reducer.world == 1 or first is None or not first.is_cuda or dist.get_backend(self.reducer.pg) != "nccl"
                or os.environ.get("THEIA_DP_RESERVED_CUS") or os.environ.get("THEIA_RCCL_MAX_NCHANNELS")):
            return {}
        dev = first.device
        results = {}
        for r in candidates:
            self._reserve = int(r)
            step_fn()  # one untimed step under the new budget
            torch.cuda.synchronize(dev)
            dist.barrier(group=self.reducer.pg)
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            torch.cuda.synchronize(dev)
            tt = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.reducer.pg)
            results[int(r)] = float(tt.item())
        self._reserve = pick_reservation(results, min_gain)
        # ... and the other way of living with collectives that hold CUs: nothing reserved, the persistent NT GEMM on its work-conserving
        # tile schedule (theia_set_gemm_schedule; alone on the chip it costs 3-8 % of a launch, so it has to earn its place on this job's
        # own step like a reservation does).  Kept only if it beats the best static arrangement by more than the margin.
        from . import ops
        self.autotune_dynamic_ms = None
        if try_dynamic and hasattr(ops, "set_gemm_schedule"):
            keep = self._reserve
            self._reserve = 0
            self.dynamic_schedule = True  # (scoped to the backward window like the CU budget: _on_bucket switches it on, _finalize off)
            try:
                step_fn()
                torch.cuda.synchronize(dev)
                dist.barrier(group=self.reducer.pg)
                t0 = time.perf_counter()
                for _ in range(steps):
                    step_fn()
                torch.cuda.synchronize(dev)
                tt = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.reducer.pg)
                self.autotune_dynamic_ms = float(tt.item())
            except BaseException:
                self.dynamic_schedule = False
                self._reserve = keep  # the arrangement pick_reservation chose stays in force
                self._restore_sched()
                raise
            # it has to beat the best static arrangement by TWICE the margin a reservation needs: the decision rests on a handful of steps,
            # and a wrong "on" costs every backward pass the schedule's 3-8 % (every rank reaches the same decision: MAX-reduced timings)
            if not self.autotune_dynamic_ms < results[keep] * (1.0 - 2.0 * min_gain):
                self.dynamic_schedule = False
                self._reserve = keep
        return results

    def _restore_sched(self) -> None:
        from . import ops
        if self._saved_sched is not None:
            ops.set_gemm_schedule(self._saved_sched)
            self._saved_sched = None

    def _on_bucket(self, bucket, side_event=None) -> None:
        if not self._callback_queued:
            # runs once when the current backward pass has finished (same mechanism DDP uses)
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            self._callback_queued = True
            if self._reserve:
                self._shrink_cus()  # launches enqueued from here on leave these CUs to the collectives
            if self.dynamic_schedule and self._saved_sched is None:
                from . import ops
                self._saved_sched = ops.set_gemm_schedule(True)  # ... or draw their tiles from the per-XCD queues; forward / eval launches never do
        self.reducer.bucket_ready(bucket.flat, side_event)

    def _finalize(self) -> None:
        try:
            self.reducer.finish()
        finally:
            self._restore_cus()
            self._restore_sched()
            self._callback_queued = False

    def forward(self, *args, **kwargs):
        # a backward pass that raised never reached _finalize: do not run the next step on the reduced budget / with stale state
        self._restore_cus()
        self._restore_sched()
        if self._callback_queued:
            self._callback_queued = False
            self.reducer._pending.clear()
        return self.module(*args, **kwargs)
