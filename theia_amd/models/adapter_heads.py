"""Per-teacher light-conv adapter head: parameter container with the reference's names
(reference ``LightConvAdapterHead`` models/adapter_heads.py:232-359).

Key layout kept for checkpoint compatibility: ``pad.1`` (ConvTranspose2d 14->16), ``adapter.{0,3,6}`` (LayerNorm
[C,H,W]), ``adapter.{1,4}`` (Conv2d for 16x16 targets / ConvTranspose2d for 64x64 targets), ``adapter.8`` (Linear).
Compute is in the HIP engine; ``kind`` tells it which branch of the reference constructor this head is."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .backbones import ConvParams, LayerNormParams, LinearParams, _Holder


class _Slot(_Holder):
    """parameter-less positions of the reference's nn.Sequential (Rearrange / ReLU)."""


class LinearAdapterHead(nn.Module):
    """CLS-token distillation head: one Linear on token 0 (reference ``LinearAdapterHead`` adapter_heads.py:28-58;
    ``adapter.0`` = nn.Linear(C, Ct)).  Parameter container; compute is in the HIP engine (``kind == "cls"``)."""

    def __init__(self, source_size, target_size):
        super().__init__()
        self.source_size, self.target_size = tuple(source_size), tuple(target_size)
        self.kind = "cls"
        self.adapter = nn.ModuleDict({"0": LinearParams(int(source_size[0]), int(target_size[0]))})
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        lin = self.adapter["0"]
        bound = 1.0 / math.sqrt(lin.weight.shape[1])
        nn.init.uniform_(lin.weight, -bound, bound)
        nn.init.uniform_(lin.bias, -bound, bound)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("LinearAdapterHead holds parameters only; use the translator's forward")


class LightConvAdapterHead(nn.Module):
    def __init__(self, source_size, target_size, hidden_size_factor: float = 1.0):
        super().__init__()
        if source_size[1] != source_size[2] or target_size[1] != target_size[2]:
            raise NotImplementedError("Currently does not support non-square feature maps like source size"
                                      "{source_size} and target size {target_size}.")
        if tuple(source_size[1:]) != (14, 14):
            raise NotImplementedError("the MI355X hot path covers the 14x14 DeiT-patch16-224 source map")
        C = int(source_size[0])
        hidden = int(C * hidden_size_factor)
        if hidden != C:
            raise NotImplementedError("hidden_size_factor != 1.0 is not part of the hot path (configs/model/translator/lconv.yaml)")
        Ct, Ht = int(target_size[0]), int(target_size[1])
        self.source_size = (C, 16, 16)  # after padding, as in the reference (adapter_heads.py:290)
        self.target_size = tuple(target_size)
        self.hidden_size_factor = hidden_size_factor
        self.pad = nn.ModuleDict({"0": _Slot(), "1": ConvParams((C, C, 3, 3), C)})  # ConvTranspose2d(C, C, 3, stride 1)
        if Ht == 64:      # adapter_heads.py:304-315
            self.kind, sizes = "up64", (16, 31, 64)
        elif Ht == 16:    # adapter_heads.py:316-327
            self.kind, sizes = "same16", (16, 16, 16)
        else:
            raise NotImplementedError(f"{tuple(source_size)} to {tuple(target_size)} is not supported.")
        self.sizes = sizes
        self.adapter = nn.ModuleDict({
            "0": LayerNormParams((C, sizes[0], sizes[0])),
            "1": ConvParams((C, C, 3, 3), C),
            "2": _Slot(),
            "3": LayerNormParams((C, sizes[1], sizes[1])),
            "4": ConvParams((C, C, 3, 3), C),
            "5": _Slot(),
            "6": LayerNormParams((C, sizes[2], sizes[2])),
            "7": _Slot(),
            "8": LinearParams(C, Ct),
        })
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """PyTorch default initialisers of Conv2d / ConvTranspose2d / Linear (kaiming-uniform a=sqrt(5))."""
        def conv_like(w, b):
            fan_in = w.shape[1] * w.shape[2] * w.shape[3]  # torch uses dim 1 for both conv and conv-transpose
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(w, -bound, bound)
            nn.init.uniform_(b, -bound, bound)
        conv_like(self.pad["1"].weight, self.pad["1"].bias)
        conv_like(self.adapter["1"].weight, self.adapter["1"].bias)
        conv_like(self.adapter["4"].weight, self.adapter["4"].bias)
        lin = self.adapter["8"]
        bound = 1.0 / math.sqrt(lin.weight.shape[1])
        nn.init.uniform_(lin.weight, -bound, bound)
        nn.init.uniform_(lin.bias, -bound, bound)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("LightConvAdapterHead holds parameters only; use the translator's forward")
