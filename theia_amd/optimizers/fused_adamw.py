"""Fused AdamW over the engine's flat parameter / gradient buckets (one HIP launch per bucket segment).

Semantics of ``torch.optim.AdamW`` (the reference's optimizer, configs/training/frame_level.yaml:18-20) with the
reference's decay grouping (optimizers/utils.py:8-35).  Parameters are re-pointed into flat fp32 buffers laid out
exactly like the gradient buckets ([decay params | no-decay params] per bucket), so a step is 2 launches per bucket."""
from __future__ import annotations

from typing import Optional

import torch

from .. import engine as _engine_mod
from .. import ops


class FusedAdamW:
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        rvfm = model.module if hasattr(model, "module") else model
        self.engine = rvfm.engine
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.param_groups = [{"lr": lr}]  # lets torch LR schedulers drive `lr`
        self.state = []
        for b in self.engine.buckets:
            dev = b.params[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdamW needs the model on the GPU (call .to('cuda') first)")
            pflat = torch.zeros(b.numel, dtype=torch.float32, device=dev)
            for i, p in enumerate(b.params):
                v = pflat[b.offsets[i]:b.offsets[i] + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
            self.state.append({"p": pflat, "m": torch.zeros_like(pflat), "v": torch.zeros_like(pflat)})
        _engine_mod.PARAM_EPOCH[0] += 1

    def zero_grad(self, set_to_none: bool = True) -> None:
        for b in self.engine.buckets:
            for p in b.params:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    ops.fill_zero(p.grad)

    def step(self) -> None:
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        for b, st in zip(self.engine.buckets, self.state):
            if b.flat is None or not any(p.requires_grad and p.grad is not None for p in b.params):
                continue
            n_decay = b.decay_numel
            if n_decay > 0:
                ops.adamw_step(st["p"][:n_decay], b.flat[:n_decay], st["m"][:n_decay], st["v"][:n_decay], lr, b1, b2, self.eps,
                               self.weight_decay, self.step_count)
            if b.numel > n_decay:
                ops.adamw_step(st["p"][n_decay:], b.flat[n_decay:], st["m"][n_decay:], st["v"][n_decay:], lr, b1, b2, self.eps, 0.0,
                               self.step_count)
        _engine_mod.PARAM_EPOCH[0] += 1
