"""Token selection / pooling of the student feature (reference: models/utils.py:8-43).

On a GPU tensor produced by the engine this dispatches to the ``theia_token_select`` HIP kernel; the function keeps
the reference's name, arguments and error behaviour."""
from typing import Optional

import torch

_MODES = {None: 0, "mean_pooling": 1, "max_pooling": 2, "cls": 3}


def handle_feature_output(x: torch.Tensor, feature_reduce_method: Optional[str] = None, num_discard_tokens: int = 0) -> torch.Tensor:
    if feature_reduce_method == "identity":
        return x
    if feature_reduce_method not in _MODES:
        raise NotImplementedError(f"feature_reduce_method {feature_reduce_method} it not implemented.")
    from .. import ops
    if not x.is_cuda:
        raise RuntimeError("theia_amd.handle_feature_output runs on the GPU only (no CPU fallback)")
    b, n, D = x.shape
    xc = x.contiguous()
    return ops.token_select(xc, b, n, D, num_discard_tokens, _MODES[feature_reduce_method])
