"""RobotVisionFM -- drop-in for the reference's ``theia.models.rvfm.RobotVisionFM`` (models/rvfm.py:15-185) whose
forward / backward run in hand-written HIP kernels on MI355X.

Same constructor signature, attributes, methods (``forward_feature``, ``forward``, ``get_loss``,
``load_pretrained_weights``, ``freeze_translator``) and state_dict keys as the reference; one extra keyword,
``precision`` ("fp32": exact-f32 MFMA path, reference-parity mode and the default; "bf16": bf16 MFMA operands with
f32 accumulation and f32 master weights, the throughput mode; "fp8": the bf16 mode with OCP e4m3 operands, per-tensor
delayed scaling, for every forward / data-gradient GEMM -- BASELINE configs[3]).
"""
from __future__ import annotations

import os
from typing import Any, Optional

import torch
import torch.nn as nn

from ..engine import StudentEngine
from .backbones import build_backbone, remap_legacy_key
from .feature_translators import build_feature_translator
from .utils import handle_feature_output


def _to_plain_dict(x):
    """translator_kwargs may be an OmegaConf node (reference: OmegaConf.to_container, rvfm.py:65) or a dict."""
    try:  # pragma: no cover - omegaconf is optional
        from omegaconf import OmegaConf  # type: ignore
        if OmegaConf.is_config(x):
            return OmegaConf.to_container(x)
    except Exception:
        pass
    return dict(x)


class RobotVisionFM(nn.Module):
    def __init__(
        self,
        backbone: str | nn.Module = "facebook/deit-small-patch16-224",
        pretrained: bool = False,
        translator: str | nn.Module = "lconv",
        target_feature_sizes: Optional[dict] = None,
        translator_kwargs: Optional[dict] = None,
        target_loss_weights: Optional[dict] = None,
        checkpoint_path: Optional[str] = None,
        feature_reduce_method: Optional[str] = None,
        image_size: int = 224,
        precision: Optional[str] = None,
        **kwargs: Any,
    ) -> None:
        super().__init__()
        self.target_feature_sizes = target_feature_sizes
        self.preprocessor = None
        self.pretrained = pretrained

        self.image_size = image_size
        if isinstance(backbone, nn.Module):
            raise NotImplementedError("pass the backbone by name; the MI355X engine owns the student's compute")
        self.backbone: nn.Module = build_backbone(backbone, pretrained, image_size=image_size, **kwargs)
        self.final_spatial = None
        self.feature_reduce_method = feature_reduce_method
        self.no_cls = hasattr(self.backbone, "no_cls")
        self.num_reg_tokens = self.backbone.num_reg_tokens if hasattr(self.backbone, "num_reg_tokens") else 0

        backbone_feature_size = self.backbone.get_feature_size(keep_spatial=True)
        if self.target_feature_sizes:
            translator_kwargs = {} if translator_kwargs is None else _to_plain_dict(translator_kwargs)
            translator_kwargs["backbone_feature_size"] = backbone_feature_size
            translator_kwargs["target_feature_sizes"] = target_feature_sizes
            if isinstance(translator, nn.Module):
                raise NotImplementedError("pass the translator by type name ('lconv')")
            self.translator = build_feature_translator(translator, **translator_kwargs)

        # the reference multiplies a tensor by this dict when it is not None (rvfm.py:159) -- only None can work
        if target_loss_weights is not None:
            raise NotImplementedError("target_loss_weights must be None (the reference's non-null path is broken, "
                                      "models/rvfm.py:159); even weights 1/T are used")
        self.target_loss_weights = None

        precision = precision or os.environ.get("THEIA_PRECISION", "fp32")
        # the engine is not a Module: it only references this model's parameters
        object.__setattr__(self, "_engine", StudentEngine(self, precision))
        self.backbone._engine = self._engine
        if hasattr(self, "translator"):
            self.translator._engine = self._engine
        if checkpoint_path:
            self.load_pretrained_weights(checkpoint_path)

    @property
    def engine(self) -> StudentEngine:
        return self._engine

    @property
    def precision(self) -> str:
        return self._engine.precision

    # ---------------------------------------------------------------- checkpoints (rvfm.py:77-92)
    def load_pretrained_weights(self, checkpoint_path: str) -> None:
        if checkpoint_path:
            weights_dict = torch.load(checkpoint_path, map_location="cpu")
            own = self.state_dict()
            pretrained_dict = {}
            for k, v in weights_dict.items():
                k2 = k if k in own else remap_legacy_key(k)  # transformers-4.4x-era key names
                if k2 in own:
                    pretrained_dict[k2] = v
            self.load_state_dict(pretrained_dict, strict=False)

    def freeze_translator(self) -> None:
        for param in self.translator.parameters():
            param.requires_grad = False

    # ---------------------------------------------------------------- forward paths (rvfm.py:94-136)
    def forward_feature(self, x: Any, **kwargs: Any) -> torch.Tensor:
        feature = self.backbone(x, **kwargs)
        return handle_feature_output(feature, feature_reduce_method=self.feature_reduce_method,
                                     num_discard_tokens=self.num_reg_tokens)

    def forward_feature_streamed(self, x: torch.Tensor, chunk: int = 512, out: Optional[torch.Tensor] = None, **kwargs: Any) -> torch.Tensor:
        """``forward_feature`` for a large uint8 batch (device-resident or pinned host memory) that is streamed through ONE
        hipGraph capture of the forward of ``chunk`` images (theia_amd/streaming.py): same values as ``forward_feature``, one
        host call per chunk.  Inference only (no autograd graph)."""
        from ..streaming import StreamedForwardFeature
        key = (int(chunk), tuple(sorted(kwargs.items())))
        sff = self.__dict__.setdefault("_streamers", {}).get(key)
        if sff is None:
            sff = self.__dict__["_streamers"][key] = StreamedForwardFeature(self, chunk, **kwargs)
        return sff(x, out=out)

    def forward(self, x: Any, target_model_names: Optional[list] = None, **kwargs: Any) -> dict:
        x = self.backbone(x, **kwargs)
        # reference: `x = x[:, :-num_reg_tokens]` here (rvfm.py:133-134).  The engine's heads read the 196 patch rows of the
        # full token matrix in place (row-map offset/stride), so the register tokens are skipped without a copy and their
        # rows of the gradient stay zero -- same values, no slice.
        return self.translator(x, target_model_names, backbone_no_cls=self.no_cls)

    # ---------------------------------------------------------------- losses (rvfm.py:138-185)
    def get_loss(self, pred_features: dict, y: dict, as_float: bool = True) -> dict:
        """Same keys as the reference.  Each teacher's three losses come from ONE fused HIP reduction; with
        ``as_float=True`` (reference behaviour: ``.item()`` per value) the per-model numbers are fetched with a single
        device->host copy; ``as_float=False`` keeps them as 0-d device tensors (no host sync in the training loop)."""
        T = len(pred_features)
        per = []
        names = list(pred_features.keys())
        for t in names:
            per.append(self._engine.distill_loss(pred_features[t], y[t], head=t))  # f32[3] = (mse, cos, l1)
        allv = torch.stack(per, 0)  # [T, 3]
        avg = allv.sum(0) * (1.0 / T)
        if as_float:
            host = allv.detach().cpu()
            mse_pm = {t: float(host[i, 0]) for i, t in enumerate(names)}
            cos_pm = {t: float(host[i, 1]) for i, t in enumerate(names)}
            l1_pm = {t: float(host[i, 2]) for i, t in enumerate(names)}
        else:
            d = allv.detach()
            mse_pm = {t: d[i, 0] for i, t in enumerate(names)}
            cos_pm = {t: d[i, 1] for i, t in enumerate(names)}
            l1_pm = {t: d[i, 2] for i, t in enumerate(names)}
        return {
            "mse_loss": avg[0], "cos_loss": avg[1], "l1_loss": avg[2],
            "mse_losses_per_model": mse_pm, "cos_losses_per_model": cos_pm, "l1_losses_per_model": l1_pm,
        }
