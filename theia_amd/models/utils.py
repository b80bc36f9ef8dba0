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
    mode = _MODES[feature_reduce_method]
    if torch.is_grad_enabled() and x.requires_grad:  # fine-tuning through forward_feature: keep the graph (reference outputs
        return _TokenSelect.apply(xc, num_discard_tokens, mode)  # are differentiable slices / means / maxima)
    return ops.token_select(xc, b, n, D, num_discard_tokens, mode)


class _TokenSelect(torch.autograd.Function):
    """theia_token_select with its backward (a scatter / broadcast into the token gradient).  The output is f32 whatever the
    compute dtype (the reference returns fp32 features); the gradient is returned in x's dtype."""

    @staticmethod
    def forward(ctx, x, disc, mode):
        from .. import ops
        b, n, D = x.shape
        out = ops.token_select(x, b, n, D, disc, mode)
        ctx.disc, ctx.mode = disc, mode
        ctx.save_for_backward(x, out) if mode == 2 else ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        b, n, D = x.shape
        hi = n - ctx.disc
        dx = torch.zeros_like(x)
        g = g.to(x.dtype)
        if ctx.mode == 0:
            dx[:, 1:hi] = g
        elif ctx.mode == 1:
            dx[:, 1:hi] = (g / (hi - 1)).unsqueeze(1)
        elif ctx.mode == 2:  # gradient to the (first) arg-max token of every channel, as torch.amax's backward splits ties evenly
            sel = x[:, 1:hi].float() == ctx.saved_tensors[1].unsqueeze(1)
            dx[:, 1:hi] = (sel / sel.sum(1, keepdim=True).clamp_min(1)).to(x.dtype) * g.unsqueeze(1)
        else:
            dx[:, 0] = g
        return dx, None, None
