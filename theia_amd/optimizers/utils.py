"""Weight-decay parameter grouping (reference optimizers/utils.py:8-35).

Rule preserved exactly: no decay for ``param.ndim <= 1`` or names ending in ``.bias``; everything else decays -- which
includes ``cls_token``, ``position_embeddings`` and the 3-D LayerNorm ``weight`` [C,H,W] of the translator heads."""
from typing import Any, Iterable

import torch.nn as nn


def is_no_decay(name: str, param) -> bool:
    return param.ndim <= 1 or name.endswith(".bias")


def param_groups_weight_decay(model: nn.Module, weight_decay: float = 1e-5,
                              no_weight_decay_parameters: Iterable[str] = ()) -> list[dict[str, Any]]:
    no_weight_decay_parameters = set(no_weight_decay_parameters)
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if is_no_decay(name, param) or name in no_weight_decay_parameters:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]
