"""HF-hub style loading of Theia checkpoints (reference README.md:22-38):

    model = AutoModel.from_pretrained("theaiinstitute/theia-base-patch16-224-cdiv", trust_remote_code=True)
    feature = model.forward_feature(images_uint8)        # student feature
    predicted = model(images_uint8)                      # dict teacher -> predicted feature

The hub repositories hold a ``config.json`` plus a weight file whose tensors are the ``RobotVisionFM`` state_dict
(``backbone.model.*``, ``translator.translator_heads.*``); their ``trust_remote_code`` module is not part of the reference
tree and nothing can be downloaded here, so this module provides the equivalent for a LOCAL snapshot directory:

    from theia_amd.hub import TheiaModel
    model = TheiaModel.from_pretrained("/path/to/theia-base-patch16-224-cdiv", feature_reduce_method=None, precision="bf16")

and ``register_with_transformers()`` hooks the ``theia`` model type into ``AutoConfig`` / ``AutoModel`` so that
``AutoModel.from_pretrained(local_dir)`` returns the same object.  Weights are filtered exactly like
``RobotVisionFM.load_pretrained_weights`` (models/rvfm.py:77-87: keys that exist in the model) after mapping
transformers-4.4x-era key names to the current ones (``remap_legacy_key``).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Optional

import torch

from .foundation_models.common import get_model_feature_size
from .models.backbones import remap_legacy_key
from .models.rvfm import RobotVisionFM

CONFIG_NAME = "config.json"
WEIGHT_NAMES = ("model.safetensors", "pytorch_model.bin", "model.pth")


def _read_weights(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)  # hub-sourced pickle: tensors only


def _target_sizes(cfg: Dict[str, Any]) -> Optional[Dict[str, tuple]]:
    """teacher -> (C, H, W) from either an explicit ``target_feature_sizes`` mapping or a list of teacher names"""
    if cfg.get("target_feature_sizes"):
        return {k: tuple(int(x) for x in v) for k, v in cfg["target_feature_sizes"].items()}
    names = cfg.get("target_model_names") or cfg.get("target_models")
    if names:
        return {t: (get_model_feature_size(t[:-4], keep_spatial=True)[:1] if t.endswith("_cls") else get_model_feature_size(t, keep_spatial=True))
                for t in names}
    return None


class TheiaModel(RobotVisionFM):
    """``RobotVisionFM`` with ``from_pretrained`` / ``save_pretrained`` on a hub-layout directory."""

    config: Dict[str, Any]

    @classmethod
    def from_pretrained(cls, path: str, device: Optional[str] = None, **overrides: Any) -> "TheiaModel":
        """path: directory with config.json + one of model.safetensors / pytorch_model.bin / model.pth (or the weight file itself,
        config.json beside it).  overrides (feature_reduce_method=, precision=, processor=, ...) replace config entries, like the
        keyword arguments of ``AutoModel.from_pretrained`` do for the hub model."""
        wfile = None
        if os.path.isfile(path):
            wfile, path = path, os.path.dirname(path)
        cfg_file = os.path.join(path, CONFIG_NAME)
        if not os.path.isfile(cfg_file):
            raise FileNotFoundError(f"{cfg_file} not found (a local snapshot of the hub repository is needed: nothing is downloaded)")
        cfg = json.load(open(cfg_file))
        cfg.update(overrides)
        if wfile is None:
            for n in WEIGHT_NAMES:
                if os.path.isfile(os.path.join(path, n)):
                    wfile = os.path.join(path, n)
                    break
        if wfile is None:
            raise FileNotFoundError(f"no weight file ({', '.join(WEIGHT_NAMES)}) in {path}")
        extra = {k: cfg[k] for k in ("processor", "num_reg_tokens") if k in cfg}
        model = cls(backbone=cfg.get("backbone", "facebook/deit-small-patch16-224"), pretrained=False,
                    translator=cfg.get("translator", "lconv"), target_feature_sizes=_target_sizes(cfg),
                    translator_kwargs=cfg.get("translator_kwargs", {"hidden_size_factor": 1.0}),
                    feature_reduce_method=cfg.get("feature_reduce_method", cfg.get("feature_reduction_method")),
                    image_size=cfg.get("image_size", 224), precision=cfg.get("precision"), **extra)
        model.config = cfg
        weights = _read_weights(wfile)
        own = model.state_dict()
        picked = {}
        for k, v in weights.items():
            k2 = k if k in own else remap_legacy_key(k)
            if k2 in own:  # rvfm.py:84-86: keep the keys the model has
                picked[k2] = v
        missing = [k for k in own if k not in picked]
        # a snapshot whose keys map to nothing (another prefix, an unknown legacy naming) would otherwise leave a randomly initialised
        # model behind without a word
        if not picked:
            raise RuntimeError(f"{wfile}: none of its {len(weights)} tensors matches a parameter of the model "
                               f"(first keys: {list(weights)[:3]}; expected e.g. {list(own)[:2]})")
        missing_backbone = [k for k in missing if k.startswith("backbone.")]
        if missing_backbone:
            import warnings
            warnings.warn(f"{wfile}: {len(missing_backbone)} backbone tensors are not in the checkpoint and keep their initial values "
                          f"(e.g. {missing_backbone[:3]})")
        model.load_state_dict(picked, strict=False)
        model.loading_info = {"loaded": len(picked), "missing_keys": missing, "unexpected_keys": len(weights) - len(picked)}
        return model.to(device) if device else model

    def save_pretrained(self, path: str, safe_serialization: bool = True) -> None:
        os.makedirs(path, exist_ok=True)
        cfg = dict(getattr(self, "config", {}))
        cfg.update({"model_type": "theia", "backbone": self.backbone.model_name, "translator": "lconv",
                    "target_feature_sizes": {k: list(v) for k, v in (self.target_feature_sizes or {}).items()},
                    "feature_reduce_method": self.feature_reduce_method, "image_size": self.image_size})
        json.dump(cfg, open(os.path.join(path, CONFIG_NAME), "w"), indent=1)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(path, "model.safetensors"))
        else:
            torch.save(sd, os.path.join(path, "pytorch_model.bin"))


def register_with_transformers() -> None:
    """``AutoModel.from_pretrained(local_dir)`` / ``AutoConfig.from_pretrained(local_dir)`` for ``"model_type": "theia"``."""
    from transformers import AutoConfig, AutoModel, PretrainedConfig

    class TheiaConfig(PretrainedConfig):
        model_type = "theia"

    class _AutoTheia:  # AutoModel only needs from_pretrained / a config class on the registered model class
        config_class = TheiaConfig

        @classmethod
        def from_pretrained(cls, path, *a, **kw):
            kw.pop("config", None)
            kw.pop("trust_remote_code", None)
            return TheiaModel.from_pretrained(path, **kw)

    try:
        AutoConfig.register("theia", TheiaConfig)
        AutoModel.register(TheiaConfig, _AutoTheia)
    except ValueError:  # already registered
        pass
