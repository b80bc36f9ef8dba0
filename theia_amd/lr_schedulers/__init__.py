"""LR schedules of the reference (lr_schedulers/lr_schedulers.py:8-77): linear warm-up then constant (default) or
cosine warm restarts.  Host-side scalar schedules: composed from torch.optim.lr_scheduler, as the reference does."""
from typing import Any

from torch.optim.lr_scheduler import ConstantLR, CosineAnnealingWarmRestarts, LinearLR, SequentialLR


def get_constant_lrs_with_linear_warm_up(optimizer, warm_up_steps: int = 2000, warm_up_lr_start_factor: float = 1e-2,
                                         warm_up_lr_end_factor: float = 1.0, **kwargs: Any) -> SequentialLR:
    warm = LinearLR(optimizer, start_factor=warm_up_lr_start_factor, end_factor=warm_up_lr_end_factor, total_iters=warm_up_steps)
    return SequentialLR(optimizer, schedulers=[warm, ConstantLR(optimizer, factor=1.0)], milestones=[warm_up_steps])


def get_cos_lrs_with_linear_warm_up(optimizer, warm_up_steps: int = 2000, warm_up_lr_start_factor: float = 1e-2,
                                    warm_up_lr_end_factor: float = 1.0, cos_lrs_T_0: int = 5000) -> SequentialLR:
    warm = LinearLR(optimizer, start_factor=warm_up_lr_start_factor, end_factor=warm_up_lr_end_factor, total_iters=warm_up_steps)
    cos = CosineAnnealingWarmRestarts(optimizer, T_0=cos_lrs_T_0, T_mult=1)
    return SequentialLR(optimizer, schedulers=[warm, cos], milestones=[warm_up_steps])
