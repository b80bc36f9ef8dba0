"""Hydra-compatible configuration for ``train_rvfm`` without requiring hydra/omegaconf (neither is installed here).

The reference composes ``configs/train_rvfm_imagenet.yaml`` (defaults: dataset=imagenet, model/backbone=deit,
model/translator=lconv, training=frame_level [-> target_models=cdiv], logging=default) and accepts CLI overrides such
as ``training/target_models=cddsv model.backbone.backbone=facebook/deit-tiny-patch16-224 training.batch_size=128``
(README.md:88).  This module reproduces that tree -- same group names, keys and default values -- as Python data, applies
``group=option`` and ``a.b.c=value`` overrides in order, and returns an attribute-access dict.  If a directory of
YAML files with the same layout is given (``--config-path``), its files take precedence over the built-in tree.
"""
from __future__ import annotations

import copy
import os
import re
from typing import Any, Dict, List

import yaml

# PyYAML implements YAML 1.1, whose float needs a dot: "2e-3" / "1e-2" (the notation of the reference's own
# configs/training/frame_level.yaml, and of README-style overrides) load as STRINGS.  Hydra/OmegaConf read them as floats.
_EXP_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _coerce(x):
    """exponent-notation strings -> float, recursively through dicts and lists"""
    if isinstance(x, str) and _EXP_FLOAT.match(x.strip()):
        return float(x)
    if isinstance(x, dict):
        return {k: _coerce(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_coerce(v) for v in x]
    return x

_TEACHER_SETS = {
    "dinov2": ["facebook/dinov2-large"],
    "vit": ["google/vit-huge-patch14-224-in21k"],
    "clip": ["openai/clip-vit-large-patch14"],
    "sam": ["facebook/sam-vit-huge"],
    "depth_anything": ["LiheYoung/depth-anything-large-hf"],
    "cdiv": ["google/vit-huge-patch14-224-in21k", "facebook/dinov2-large", "openai/clip-vit-large-patch14"],
    "cddsv": ["google/vit-huge-patch14-224-in21k", "facebook/dinov2-large", "openai/clip-vit-large-patch14",
              "facebook/sam-vit-huge", "LiheYoung/depth-anything-large-hf"],
}

GROUPS: Dict[str, Dict[str, Any]] = {
    "dataset": {
        "imagenet": {"return_metadata": False, "shuffle": True, "shuffle_buffer_size": 1024, "feature_norm": True,
                     "dataset_root": "/storage/nfs/datasets/jshang/", "dataset_ratio": 0.1, "load_action": False,
                     "dataset_mix": ["imagenet"]},
        # MI355X build addition: synthetic uint8 images + random teacher features of the right shapes (no I/O)
        "synthetic": {"return_metadata": False, "shuffle": False, "shuffle_buffer_size": 0, "feature_norm": False,
                      "dataset_root": "", "dataset_ratio": 1.0, "load_action": False, "dataset_mix": ["synthetic"],
                      "train_steps_per_epoch": 20, "eval_steps_per_epoch": 2},
    },
    "model/backbone": {"deit": {"backbone": "facebook/deit-small-patch16-224", "pretrained": False}},
    "model/translator": {"lconv": {"type": "lconv", "kwargs": {"hidden_size_factor": 1.0}}},
    "training": {
        "frame_level": {
            "epochs": 50, "warm_up_steps_ratio": 0.1, "base_lr": 2e-3, "batch_size": 16, "random_target_models": -1,
            "num_workers": 8, "base_batch_size": 64, "base_world_size": 8, "weight_decay": 0.01,
            "optimizer": {"_target_": "torch.optim.AdamW", "betas": [0.9, 0.999]},
            "lr_scheduler": {"_target_": "theia.lr_schedulers.get_constant_lrs_with_linear_warm_up", "warm_up_lr_start_factor": 1e-2},
            "grad_clip": False, "grad_clip_norm_warmup": 10.0, "grad_clip_norm": 1.0,
            "freeze_translator": False, "freeze_translator_start_steps_ratio": 0.2, "translator_lr_factor": 1.0,
            "main_loss": "cos_l1",
            "__defaults__": {"training/target_models": "cdiv"},
        }
    },
    "training/target_models": {k: {"target_model_names": v, "target_model_weights": None} for k, v in _TEACHER_SETS.items()},
    "logging": {"default": {"model_path": "./trained_models", "log_path": "./logs", "save_ckpt_interval": 20000, "notes": "",
                            "run_identifier_prefix": "", "project": "theia"}},
}
ROOT_DEFAULTS = [("dataset", "imagenet"), ("model/backbone", "deit"), ("model/translator", "lconv"), ("training", "frame_level"),
                 ("logging", "default")]
ROOT_VALUES = {"seed": 0}


class Cfg(dict):
    """dict with attribute access (enough of DictConfig for the training script)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _load_group(group: str, option: str, config_path: str | None) -> Dict[str, Any]:
    if config_path:
        f = os.path.join(config_path, group, option + ".yaml")
        if os.path.exists(f):
            d = _coerce(yaml.safe_load(open(f)) or {})
            dd = {k: v for k, v in d.items() if k != "defaults"}
            nested = {}
            for item in d.get("defaults", []) or []:
                if isinstance(item, dict):
                    for g, o in item.items():
                        nested[f"{group}/{g}"] = o
                elif isinstance(item, str) and item != "_self_":  # same-group base config, e.g. image_video_default
                    base = _load_group(group, item, config_path)
                    nested.update(base.pop("__defaults__", {}))
                    base.update(dd)
                    dd = base
            if nested:
                dd["__defaults__"] = nested
            return dd
    try:
        return copy.deepcopy(GROUPS[group][option])
    except KeyError as e:
        raise KeyError(f"unknown config option {group}={option}; known: {sorted(GROUPS.get(group, {}))}") from e


def _place(root: Dict[str, Any], group: str, value: Dict[str, Any]) -> None:
    node = root
    parts = group.split("/")
    for p in parts[:-1]:
        node = node.setdefault(p, {})
    node.setdefault(parts[-1], {}).update(value)


def _parse_value(s: str):
    try:
        return _coerce(yaml.safe_load(s))
    except yaml.YAMLError:
        return s


def compose(overrides: List[str] | None = None, config_path: str | None = None) -> Cfg:
    overrides = list(overrides or [])
    choices = dict(ROOT_DEFAULTS)
    group_names = set(GROUPS.keys())
    value_overrides = []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        if k in group_names or "/" in k:
            choices[k] = v
        else:
            value_overrides.append((k, _parse_value(v)))
    root: Dict[str, Any] = copy.deepcopy(ROOT_VALUES)
    pending = list(ROOT_DEFAULTS)
    pending = [(g, choices.get(g, o)) for g, o in pending]
    seen = set()
    while pending:
        g, o = pending.pop(0)
        if g in seen:
            continue
        seen.add(g)
        val = _load_group(g, o, config_path)
        nested = val.pop("__defaults__", {})
        _place(root, g, val)
        for ng, no in nested.items():
            pending.append((ng, choices.get(ng, no)))
    for k, v in value_overrides:
        node = root
        parts = k.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], dict):
                node[p] = {}
            node = node[p]
        node[parts[-1]] = v
    return _wrap(root)


def instantiate(spec: Dict[str, Any], *args, **kwargs):
    """hydra.utils.instantiate for the two ``_target_`` uses of the training config (optimizer, lr scheduler).
    ``theia.*`` targets resolve to this package."""
    import importlib
    spec = dict(spec)
    target = spec.pop("_target_")
    if target.startswith("theia."):
        target = "theia_amd." + target[len("theia."):]
    mod, name = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(mod), name)
    conv = {k: (tuple(v) if isinstance(v, list) else v) for k, v in spec.items()}
    conv.update(kwargs)
    return fn(*args, **conv)


def to_yaml(cfg) -> str:
    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, list):
            return [plain(v) for v in x]
        return x
    return yaml.safe_dump(plain(cfg), sort_keys=False)
