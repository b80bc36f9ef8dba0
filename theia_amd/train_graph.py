"""One whole train step -- zero_grad, student forward, translator heads, distillation losses, backward, (gradient clipping,) fused AdamW
and the operand rebuild -- captured ONCE into a hipGraph and replayed per step.

Reference loop: scripts/train/train_rvfm.py:116-131 (``pred = rvfm(images); losses = rvfm.module.get_loss(pred, targets);
main_loss.backward(); clip_grad_norm_; optimizer.step()``).  Eagerly that is ~1500 C-ABI launches per step through ctypes; the host
needs 11-14 ms to enqueue them.  At the headline configuration (DeiT-base, per-GPU batch 128: 43 ms of GPU time) the enqueue hides
behind the GPU, but the reference's DEFAULT per-GPU batch is 16 (configs/training/frame_level.yaml:8) and DeiT-tiny's whole step
at batch 256 is ~15 ms: there the drop-in is host-bound.  Stream capture of the very same launches (no tracing, no re-compilation;
single-stream: the weight-gradient side queue runs inline inside the capture) makes a step ONE host call.

What has to live on the device for that: the per-step scalars of the optimizer (learning rate from the scheduler, the two Adam bias
corrections) -- ``theia_adamw_step_dev`` reads them from a 3-float device tensor that ``FusedAdamW.prepare_step()`` refreshes before
every replay; the clip coefficient already was device-resident.  Inputs are copied into static buffers in front of the replay;
the losses come back as 0-d device tensors (reading them is the caller's synchronisation, as with ``get_loss(as_float=False)``).

World size > 1 (round 6): the step is captured as TWO halves with the gradient exchange between them --
    graph A  zero_grad, forward, losses, backward (every gradient bucket complete; the engine's per-bucket hook is off inside)
    eager    the bucket exchange of theia_amd/parallel.py (RCCL, the same ``GradBucketReducer`` calls as the plain loop, in bucket order)
    graph B  clipping, fused AdamW, operand rebuild
-- three host calls + one collective per bucket instead of ~1500 launches.  The exchange does not overlap the backward pass in this mode; it
is for the host-bound regime the capture exists for (the reference's default per-GPU batch of 16, train_rvfm.py:211-213,258 with
frame_level.yaml:8: a DeiT-small / cdiv step is ~7.5 ms of GPU time and ~100 MB of gradients, ~1 ms over xGMI).  ``split=True`` forces the
two-half form in a single process (what the bit-identity test runs: the exchange is then a no-op).
fp8 mode (host-side calibration of the first use of every scale slot) is not capturable.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import torch


def default_main_loss(losses: Dict[str, Any]) -> torch.Tensor:
    """the reference's default objective (train_rvfm.py:119-122 with main_loss = cos_l1)"""
    return 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]


class CapturedTrainStep:
    """``step = CapturedTrainStep(model, optimizer); losses = step(images_uint8, targets)``

    model:      ``RobotVisionFM`` on the GPU (bf16 or fp32 precision);  optimizer: ``FusedAdamW`` over it
    main_loss:  losses dict -> scalar tensor (default 0.9 cos + 0.1 smooth-L1);  grad_clip: max norm or None
    warmup:     the first ``warmup`` calls run eagerly (they are real training steps: operand cache, workspaces, kernel attributes and
                allocator pools exist before the capture); call ``warmup + 1`` captures and replays
    Returns the dict of ``get_loss(..., as_float=False)`` plus ``"main_loss"`` (and ``"grad_norm"`` with clipping): device tensors that
    are overwritten by the next call.  A change of input shapes re-captures."""

    def __init__(self, model, optimizer, main_loss: Callable = default_main_loss, grad_clip: Optional[float] = None, warmup: int = 2,
                 split: Optional[bool] = None, reducer=None):
        """model: the ``RobotVisionFM``, or the ``TheiaDataParallel`` wrapper around it (world size > 1: its reducer exchanges the buckets
        between the two captured halves; ``reducer`` passes one explicitly).  split: None = two halves iff a reducer with world > 1."""
        if hasattr(model, "reducer") and hasattr(model, "module"):  # TheiaDataParallel
            reducer = reducer if reducer is not None else model.reducer
            model = model.module
        import torch.distributed as dist
        if reducer is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise ValueError("CapturedTrainStep at world size > 1 needs the TheiaDataParallel wrapper (or its reducer): the gradient "
                             "buckets are exchanged between the two captured halves of the step")
        self.reducer = reducer
        self.split = bool(split) if split is not None else (reducer is not None and reducer.world > 1)
        self._graph_b = None
        if getattr(model, "precision", None) == "fp8":
            raise NotImplementedError("CapturedTrainStep: fp8 mode calibrates its scale slots from the host and is not capturable")
        self.model, self.opt, self.main_loss, self.grad_clip, self.warmup = model, optimizer, main_loss, grad_clip, max(1, int(warmup))
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("CapturedTrainStep runs on a ROCm GPU only (no CPU fallback)")
        optimizer.enable_capturable()
        self.stream = torch.cuda.Stream(device=self.device)
        self.calls = 0
        self.replays = 0
        self._key = None
        self._graph = None
        self._x = None
        self._y: Dict[str, torch.Tensor] = {}
        self._out: Dict[str, Any] = {}

    def set_grad_clip(self, max_norm: Optional[float]) -> None:
        """change the clipping threshold (train_rvfm.py:126-130 switches from grad_clip_norm_warmup to grad_clip_norm after the warm-up
        steps): the threshold is a launch argument of the captured clip kernel, so a new value voids the capture"""
        if max_norm != self.grad_clip:
            self.grad_clip = max_norm
            self._graph = None

    def invalidate(self) -> None:
        """void the capture (e.g. after ``freeze_translator()``: which parameters have gradients is baked into the captured step)"""
        self._graph = None

    # ------------------------------------------------------------------ the two halves (world size > 1, or split=True)
    def _body_a(self) -> Dict[str, Any]:
        """zero_grad .. backward.  The engine's per-bucket hook (TheiaDataParallel: start the bucket's exchange on the side stream) must
        not fire inside a capture: the buckets are exchanged behind this half, eagerly."""
        eng = self.model.engine
        hook, eng.bucket_ready_hook = eng.bucket_ready_hook, None
        try:
            self.opt.zero_grad(set_to_none=True)
            pred = self.model(self._x)
            losses = self.model.get_loss(pred, self._y, as_float=False)
            main = self.main_loss(losses)
            main.backward()
        finally:
            eng.bucket_ready_hook = hook
        out = dict(losses)
        out["main_loss"] = main.detach()
        return out

    def _exchange(self) -> None:
        """every gradient bucket through the reducer (backward-completion order), then wait on this stream -- eager, every step"""
        if self.reducer is None or self.reducer.world == 1:
            return
        for b in self.model.engine.buckets:
            if b.flat is not None:
                self.reducer.bucket_ready(b.flat)
        self.reducer.finish()

    def _body_b(self, out: Dict[str, Any]) -> None:
        if self.grad_clip is not None:
            out["grad_norm"] = self.opt.clip_grad_norm_(self.grad_clip)
        self.opt.step()

    # ------------------------------------------------------------------ the step itself (eager and captured: the same code)
    def _body(self) -> Dict[str, Any]:
        self.opt.zero_grad(set_to_none=True)
        pred = self.model(self._x)
        losses = self.model.get_loss(pred, self._y, as_float=False)
        main = self.main_loss(losses)
        main.backward()
        out = dict(losses)
        out["main_loss"] = main.detach()
        if self.grad_clip is not None:
            out["grad_norm"] = self.opt.clip_grad_norm_(self.grad_clip)
        self.opt.step()
        return out

    def static_inputs(self, images: torch.Tensor, targets: Dict[str, torch.Tensor]):
        """The static device buffers the captured step reads, shaped like ``images`` / ``targets`` (allocated on first use; a new shape
        voids the capture).  A data pipeline that writes its batches straight into them (H2D copies from pinned memory, the feature
        ingest kernel) and then calls ``step(x_static, y_static)`` with these very tensors pays no staging copy -- at DeiT-base, 5
        teachers, batch 128 the f32 teacher features of one step are 1 GB."""
        key = (tuple(images.shape), images.dtype) + tuple((t, tuple(v.shape), v.dtype) for t, v in targets.items())
        if key != self._key:  # first call / new shapes: new static buffers, the old capture (if any) is void
            self._x = torch.empty_like(images, device=self.device)
            self._y = {t: torch.empty_like(v, device=self.device) for t, v in targets.items()}
            self._key, self._graph = key, None
        return self._x, self._y

    def _stage_inputs(self, images: torch.Tensor, targets: Dict[str, torch.Tensor]) -> None:
        self.static_inputs(images, targets)
        if images is not self._x:
            self._x.copy_(images, non_blocking=True)
        for t, v in targets.items():
            if v is not self._y[t]:
                self._y[t].copy_(v, non_blocking=True)

    def _capture(self, fn):
        """fn() captured on self.stream, single-stream (see __call__); -> (graph, what fn returned)"""
        sq = getattr(self.model.engine, "_sideq", None)
        was = None
        if sq is not None:
            sq.join()
            was, sq.enabled = sq.enabled, False
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                ret = fn()
        finally:
            if sq is not None:
                sq.enabled = was
        return g, ret

    def _call_split(self) -> None:
        from . import engine as _eng
        if self.calls <= self.warmup:
            self._out = self._body_a()
            self._exchange()
            self._body_b(self._out)
            return
        if self._graph is None:
            self._graph, self._out = self._capture(self._body_a)
            self._graph_b = None
        self._graph.replay()
        self._exchange()
        if self._graph_b is None:
            # (captured after graph A has run once: the gradient buffers and the clip scratch exist; capturing B does not execute it)
            self._graph_b, _ = self._capture(lambda: self._body_b(self._out))
        self._graph_b.replay()
        self.opt._prepared = False
        self.replays += 1
        _eng.PARAM_EPOCH[0] += 1

    def __call__(self, images: torch.Tensor, targets: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        if images.dtype != torch.uint8 or images.dim() != 4:
            raise TypeError("CapturedTrainStep takes a uint8 [B, H, W, 3] / [B, 3, H, W] image batch")
        from . import engine as _eng
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        self.calls += 1
        with torch.cuda.stream(self.stream):
            self._stage_inputs(images, targets)
            self.opt.prepare_step()
            if self.split:
                self._call_split()
            elif self.calls <= self.warmup:
                self._out = self._body()
            else:
                if self._graph is None:
                    # The capture is SINGLE-STREAM: the engine's side queue (weight gradients on a second stream, joined through events) is
                    # switched to inline for the duration.  A capture that forks and joins through events replays correctly, but on ROCm 7.0
                    # destroying such a graph corrupts the host heap ("double free or corruption" within a few create / destroy cycles:
                    # tools/stress_captured_step.py; none in 80 cycles single-stream) -- and it buys nothing: the replay runs the
                    # weight-gradient branch without overlap anyway.  Same kernels in the same order: bit-identical results.
                    sq = getattr(self.model.engine, "_sideq", None)
                    was = None
                    if sq is not None:
                        sq.join()
                        was, sq.enabled = sq.enabled, False
                    try:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=self.stream):
                            self._out = self._body()
                    finally:
                        if sq is not None:
                            sq.enabled = was
                    self._graph = g
                self._graph.replay()
                self.opt._prepared = False  # (consumed by the replayed step)
                self.replays += 1
                # the replay updated the parameters behind the host's back: anything that keys on the parameter epoch (the engine's
                # operand cache in an eager call, StreamedForwardFeature's capture) must see a new one
                _eng.PARAM_EPOCH[0] += 1
        cur.wait_stream(self.stream)
        if images.is_cuda:
            images.record_stream(self.stream)
        for v in targets.values():
            if v.is_cuda:
                v.record_stream(self.stream)
        return self._out
