"""Fused AdamW over the engine's flat parameter / gradient buckets (one HIP launch per contiguous run of trainable parameters).

Semantics of ``torch.optim.AdamW`` (the reference's optimizer, configs/training/frame_level.yaml:18-20) with the
reference's decay grouping (optimizers/utils.py:8-35).  Parameters are re-pointed into flat fp32 buffers laid out
exactly like the gradient buckets ([decay params | no-decay params] per bucket), so a step is 2 launches per bucket when
every parameter of the bucket is trainable.  Parameters with ``requires_grad=False`` (``freeze_translator``, frozen
embeddings, ...) or without a gradient are left untouched -- no Adam update, no weight decay -- like torch's optimizer.

It IS a ``torch.optim.Optimizer`` (one param group holding every parameter), so the reference's LR schedulers
(``theia.lr_schedulers.*``: ``SequentialLR`` of ``LinearLR`` and ``ConstantLR`` / ``CosineAnnealingWarmRestarts``) drive
``param_groups[0]["lr"]`` exactly as they drive ``torch.optim.AdamW``."""
from __future__ import annotations

from typing import List, Tuple

import torch

from .. import engine as _engine_mod
from .. import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        rvfm = model.module if hasattr(model, "module") else model
        self.engine = rvfm.engine
        params = [p for b in self.engine.buckets for p in b.params]
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.step_count = 0
        self._clip = None  # (norm, coefficient) device pair set by clip_grad_norm_ and consumed by the next step()
        # capturable mode (theia_amd/train_graph.py): the scalars that change every step -- learning rate, the two bias corrections --
        # are read by the update kernel from `_hyper` (3 floats on the device) instead of being kernel arguments, so that a captured
        # step() can be replayed; `prepare_step()` advances the step counter and refreshes them (three fill launches outside the graph)
        self.capturable = False
        self._hyper = None
        self.flat_state = []
        for b in self.engine.buckets:
            dev = b.params[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdamW needs the model on the GPU (call .to('cuda') first)")
            pflat = torch.zeros(b.numel, dtype=torch.float32, device=dev)
            for i, p in enumerate(b.params):
                v = pflat[b.offsets[i]:b.offsets[i] + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
            self.flat_state.append({"p": pflat, "m": torch.zeros_like(pflat), "v": torch.zeros_like(pflat)})
        _engine_mod.PARAM_EPOCH[0] += 1

    # convenience mirrors of the single param group
    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._clip = None  # a coefficient computed for gradients that are being discarded must not reach a later step()
        for b in self.engine.buckets:
            for p in b.params:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    ops.fill_zero(p.grad)

    @staticmethod
    def _runs(bucket, lo: int, hi: int) -> List[Tuple[int, int]]:
        """[start, end) element ranges of the flat bucket covering maximal runs of consecutive parameters lo..hi-1 that are
        trainable and have a gradient living in the bucket"""
        runs: List[Tuple[int, int]] = []
        cur = None
        for i in range(lo, hi):
            p = bucket.params[i]
            s_i = bucket.offsets[i]
            e_i = bucket.offsets[i + 1] if i + 1 < len(bucket.params) else bucket.numel  # includes the alignment padding (zeros)
            live = p.requires_grad and p.grad is not None
            if live and p.grad.data_ptr() != bucket.flat.data_ptr() + 4 * s_i:
                bucket.view(i).copy_(p.grad)  # a foreign .grad tensor (set by user code): bring it into the bucket
            if live:
                cur = [s_i, e_i] if cur is None else [cur[0], e_i]
            elif cur is not None:
                runs.append((cur[0], cur[1]))
                cur = None
        if cur is not None:
            runs.append((cur[0], cur[1]))
        return runs

    def _live_runs(self):
        """(bucket, state, lo, hi, decay?) of every maximal range of trainable parameters with a gradient, in bucket order"""
        for b, st in zip(self.engine.buckets, self.flat_state):
            if b.flat is None:
                continue
            n_dec = sum(1 for i in range(len(b.params)) if b.offsets[i] < b.decay_numel)
            for lo, hi, decay in ((0, n_dec, True), (n_dec, len(b.params), False)):
                for s, e in self._runs(b, lo, hi):
                    yield b, st, s, e, decay

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """``nn.utils.clip_grad_norm_`` (2-norm over every parameter with a gradient; train_rvfm.py:126-130) on the flat buckets, on
        the device: partial sums of squares per live range (fixed order), one finalize launch -> total norm and the factor
        min(1, max_norm / (norm + 1e-6)).  The NEXT ``step()`` multiplies every gradient with that factor inside the AdamW kernel; the
        ``.grad`` tensors themselves stay UNSCALED (torch scales them in place: code that reads ``.grad`` between this call and
        ``step()`` sees unclipped values here); ``zero_grad()`` drops a coefficient that was not consumed; a NaN norm gives a NaN
        coefficient (every updated parameter becomes NaN, as with torch).  Returns the total norm as a 0-d device tensor, like
        torch does -- reading it is the caller's host synchronisation, not this function's."""
        from .. import _native as N
        runs = list(self._live_runs())
        dev = self.flat_state[0]["p"].device
        nb = N.lib().theia_grad_sumsq_blocks()
        out = torch.zeros(2, dtype=torch.float32, device=dev)
        if not runs:
            out[1] = 1.0
            self._clip = out
            return out[0]
        partials = torch.empty(len(runs) * nb, dtype=torch.float32, device=dev)
        for i, (b, _st, s, e, _d) in enumerate(runs):
            ops.grad_sumsq(b.flat[s:e], partials[i * nb:(i + 1) * nb])
        ops.grad_clip_coef(partials, max_norm, out)
        self._clip = out
        return out[0]

    def enable_capturable(self) -> None:
        if self._hyper is None:
            dev = self.flat_state[0]["p"].device
            self._hyper = torch.zeros(3, dtype=torch.float32, device=dev)
            self._hyper[1:3].fill_(1.0)  # bias corrections of "no step yet": never a division by zero
        self.capturable = True

    @torch.no_grad()
    def prepare_step(self) -> None:
        """capturable mode: advance the step counter and put (lr, 1 - beta1^t, 1 - beta2^t) of the step that is about to run on the
        device, on the current stream (call right before ``step()`` / before replaying a graph that contains it)"""
        self.step_count += 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        # three fill launches: the scalars travel as KERNEL ARGUMENTS (copied at launch).  An asynchronous copy from one pinned host buffer
        # raced with the host, which runs several steps ahead of the GPU and overwrote the buffer before the copy had executed -- step t
        # then used step t+k's learning rate (caught by the train-script test, which does not synchronise every step)
        self._prepared = True
        self._hyper[0:1].fill_(float(g["lr"]))
        self._hyper[1:2].fill_(1.0 - b1 ** self.step_count)
        self._hyper[2:3].fill_(1.0 - b2 ** self.step_count)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        clip, self._clip = self._clip, None
        if self.capturable:  # the per-step scalars come from the device
            # an eager step() of a capturable optimizer without prepare_step() (e.g. user code between two captured phases) refreshes
            # them itself; inside a capture they must already be in place (CapturedTrainStep calls prepare_step before every replay)
            if not getattr(self, "_prepared", False):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedAdamW.step() inside a stream capture without prepare_step(): the captured update would "
                                       "bake in stale device scalars (learning rate / bias corrections) -- call prepare_step() before "
                                       "the capture and before every replay (CapturedTrainStep does)")
                self.prepare_step()
            self._prepared = False
            for b, st, s, e, decay in self._live_runs():
                ops.adamw_step_dev(st["p"][s:e], b.flat[s:e], st["m"][s:e], st["v"][s:e], b1, b2, eps, wd if decay else 0.0, self._hyper,
                                   None if clip is None else clip[1:2])
            _engine_mod.PARAM_EPOCH[0] += 1
            return loss
        self.step_count += 1
        for b, st, s, e, decay in self._live_runs():
            if clip is None:
                ops.adamw_step(st["p"][s:e], b.flat[s:e], st["m"][s:e], st["v"][s:e], lr, b1, b2, eps, wd if decay else 0.0, self.step_count)
            else:
                ops.adamw_step_scaled(st["p"][s:e], b.flat[s:e], st["m"][s:e], st["v"][s:e], lr, b1, b2, eps, wd if decay else 0.0,
                                      self.step_count, clip[1:2])
        _engine_mod.PARAM_EPOCH[0] += 1
        return loss

    # checkpointing: flat moments + step counter (the per-parameter ``state`` dict of torch optimizers is not used)
    def _layout(self):
        """[[(parameter name, offset in the bucket's flat buffer, numel)]] per bucket -- what a checkpoint's moments are matched against"""
        return [[(n, int(o), int(p.numel())) for n, o, p in zip(b.names, b.offsets, b.params)] for b in self.engine.buckets]

    def state_dict(self):
        return {"step_count": self.step_count, "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "m": [st["m"].clone() for st in self.flat_state], "v": [st["v"].clone() for st in self.flat_state],
                "layout_version": 2, "layout": self._layout()}

    def load_state_dict(self, sd) -> None:
        """Validates BEFORE copying anything.  A checkpoint with the same per-bucket sizes loads positionally; one written under another
        bucket layout (round 3: 4 ViT buckets, round 4: 6) is re-mapped by parameter name when it carries its layout, refused otherwise."""
        ms, vs = list(sd["m"]), list(sd["v"])
        mine = [int(st["m"].numel()) for st in self.flat_state]
        if len(ms) != len(vs):
            raise ValueError("FusedAdamW.load_state_dict: 'm' and 'v' lists differ in length")
        plan = None
        lay = sd.get("layout")
        # positional load only when the layouts are known to agree: equal per-bucket sizes with another parameter order or other offsets
        # inside a bucket would attach the moments to the wrong parameters (a checkpoint without a layout -- rounds 1-3 -- can only be
        # checked by size)
        same = [int(m.numel()) for m in ms] == mine and (lay is None or [[tuple(e) for e in bk] for bk in lay] == self._layout())
        if not same:
            if lay is None:
                raise ValueError(f"FusedAdamW.load_state_dict: checkpoint buckets {[int(m.numel()) for m in ms]} do not match this model's "
                                 f"{mine} and the checkpoint carries no layout to re-map by parameter name")
            src = {name: (bi, int(off), int(n)) for bi, bucket in enumerate(lay) for name, off, n in bucket}
            plan = []
            for bi, bucket in enumerate(self._layout()):
                for name, off, n in bucket:
                    if name not in src or src[name][2] != n or src[name][1] + n > int(ms[src[name][0]].numel()):
                        raise ValueError(f"FusedAdamW.load_state_dict: parameter {name!r} ({n} elements) is not in the checkpoint's layout")
                    plan.append((bi, off, src[name][0], src[name][1], n))
            have = {name for bucket in self._layout() for name, _o, _n in bucket}
            extra = sorted(set(src) - have)
            if extra:  # moments of parameters this model does not have: refuse rather than drop them silently
                raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint holds moments for parameters this model lacks: {extra[:5]}"
                                 f"{' ...' if len(extra) > 5 else ''}")
        self.step_count = int(sd["step_count"])
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g.update(sg)
        if plan is None:
            for st, m, v in zip(self.flat_state, ms, vs):
                st["m"].copy_(m)
                st["v"].copy_(v)
        else:
            for bi, off, sb, soff, n in plan:
                self.flat_state[bi]["m"][off:off + n].copy_(ms[sb][soff:soff + n])
                self.flat_state[bi]["v"][off:off + n].copy_(vs[sb][soff:soff + n])
