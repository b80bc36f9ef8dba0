"""``lconv`` feature translator (reference models/feature_translators.py:12-88,159-205,293-313).

backbone_adapter = Identity and translator_stem = Identity for ``lconv`` (feature_translators.py:56,183), so the
translator is just the per-teacher heads; head module names are the teacher names with '.' -> '_' (:46)."""
from __future__ import annotations

from typing import Any, Optional

import torch
import torch.nn as nn

from .adapter_heads import LightConvAdapterHead, LinearAdapterHead


class LightConvFeatureTranslator(nn.Module):
    def __init__(self, backbone_feature_size, target_feature_sizes: dict, translator_hidden_size: int = 1024,
                 hidden_size_factor: float = 1.0) -> None:
        super().__init__()
        self.backbone_feature_size = backbone_feature_size
        self.target_feature_sizes = target_feature_sizes
        self.translator_hidden_size = translator_hidden_size
        self.hidden_size_factor = hidden_size_factor
        self.target_model_names = list(target_feature_sizes.keys())
        self.legit_target_model_name_map = {t: t.replace(".", "_") for t in self.target_model_names}
        heads = {}
        for t, size in target_feature_sizes.items():
            if "_cls" in t:  # feature_translators.py:193-197: "<teacher>_cls" targets get a Linear head on the CLS token
                heads[self.legit_target_model_name_map[t]] = LinearAdapterHead(backbone_feature_size, size)
            else:
                heads[self.legit_target_model_name_map[t]] = LightConvAdapterHead(backbone_feature_size, size, hidden_size_factor)
        self.translator_heads = nn.ModuleDict(heads)
        self._engine = None  # set by RobotVisionFM

    def forward(self, x: torch.Tensor, target_model_names: Optional[list] = None, backbone_no_cls: bool = False) -> dict:
        if self._engine is None:
            raise RuntimeError("the translator is driven by RobotVisionFM's engine")
        # backbone_no_cls (nocls- students: heads take all 196 tokens instead of x[:, 1:], adapter_heads.py:355-356) is a property
        # of the engine's token layout: it reads the patch rows in place for every student
        assert bool(backbone_no_cls) == (self._engine.tok0 == 0), "backbone_no_cls does not match the student's token layout"
        names = target_model_names if target_model_names is not None else self.target_model_names
        return self._engine.translator(x, list(names))


def build_feature_translator(translator_type: str, **kwargs: Any) -> nn.Module:
    if translator_type == "lconv":
        return LightConvFeatureTranslator(**kwargs)
    if translator_type in ("mlp", "conv", "transformer", "trans"):
        raise NotImplementedError(f"translator '{translator_type}' is out of the hot-path scope (only 'lconv', the reference default)")
    raise NotImplementedError(f"Requested {translator_type} is not implemented yet.")
