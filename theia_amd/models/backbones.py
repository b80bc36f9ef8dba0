"""DeiT/ViT student backbone: parameter containers with the REFERENCE state_dict layout + the engine-backed forward.

Mirrors the reference's ``models/backbones.py`` (``DeiT`` :255-341, ``build_backbone`` :506-526).  In the reference the
arithmetic lives in HuggingFace ``ViTModel`` + the HF image processor; here the modules only HOLD the fp32 master
parameters under the same names (``model.embeddings...``, ``model.layers.{i}...``, ``model.layernorm``) and every
forward/backward runs in the HIP kernels driven by ``theia_amd.engine``.
"""
from __future__ import annotations

import math
from typing import Any, Optional

import torch
import torch.nn as nn

# (hidden D, heads, mlp F) -- HF ViTConfig of facebook/deit-{tiny,small,base}-patch16-224; other fields are ViTConfig
# defaults (12 layers, patch 16, image 224, layer_norm_eps 1e-12, erf GELU, qkv_bias) -- SURVEY.md sec. 8(c).
ARCH = {
    "facebook/deit-tiny-patch16-224": (192, 3, 768),
    "facebook/deit-small-patch16-224": (384, 6, 1536),
    "facebook/deit-base-patch16-224": (768, 12, 3072),
}
NUM_LAYERS = 12
PATCH = 16
IMAGE = 224
GRID = 14
NTOK = 197          # tokens of the plain DeiT at 224x224: CLS + 14*14 patches
MAX_TOKENS = 256    # sequence length the attention kernels hold in LDS
INIT_RANGE = 0.02  # ViTConfig.initializer_range


def backbone_variant(model_name: str):
    """(base HF name, has CLS token, register tokens) -- dispatch order of the reference's build_backbone (backbones.py:519-526):
    "reg" is tested before "nocls"."""
    if "reg" in model_name:
        return model_name.replace("reg-", ""), True, None  # register count comes from the num_reg_tokens kwarg (default 7)
    if "nocls" in model_name:
        return model_name.replace("nocls-", ""), False, 0
    return model_name, True, 0


class _Holder(nn.Module):
    """A module that only owns parameters; calling it is a bug (compute lives in the engine)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} holds parameters only; use the owning model's forward")


class LinearParams(_Holder):
    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))


class LayerNormParams(_Holder):
    def __init__(self, shape):
        super().__init__()
        shape = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (shape,)
        self.weight = nn.Parameter(torch.ones(shape))
        self.bias = nn.Parameter(torch.zeros(shape))


class ConvParams(_Holder):
    """weight in PyTorch layout: Conv2d [co, ci, kh, kw]; ConvTranspose2d [ci, co, kh, kw]."""

    def __init__(self, weight_shape, bias_size: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*weight_shape))
        self.bias = nn.Parameter(torch.empty(bias_size))


class _PatchEmbeddings(_Holder):
    def __init__(self, D: int):
        super().__init__()
        self.projection = ConvParams((D, 3, PATCH, PATCH), D)


class _Embeddings(_Holder):
    """Parameter names / registration order of HF ViTEmbeddings and of the reference's NoCLS / Reg subclasses
    (backbones.py:26-37: cls_token removed, position_embeddings keeps its 197 rows; :128-150: reg_token and reg_pos_embed)."""

    def __init__(self, D: int, has_cls: bool = True, num_reg_tokens: int = 0):
        super().__init__()
        if has_cls:
            self.cls_token = nn.Parameter(torch.empty(1, 1, D))
        self.position_embeddings = nn.Parameter(torch.empty(1, NTOK, D))
        if num_reg_tokens > 0:
            self.reg_token = nn.Parameter(torch.empty(1, num_reg_tokens, D))
            self.reg_pos_embed = nn.Parameter(torch.empty(1, num_reg_tokens, D))
        self.patch_embeddings = _PatchEmbeddings(D)


class _Attention(_Holder):
    def __init__(self, D: int):
        super().__init__()
        self.q_proj = LinearParams(D, D)
        self.k_proj = LinearParams(D, D)
        self.v_proj = LinearParams(D, D)
        self.o_proj = LinearParams(D, D)


class _MLP(_Holder):
    def __init__(self, D: int, F: int):
        super().__init__()
        self.fc1 = LinearParams(D, F)
        self.fc2 = LinearParams(F, D)


class _Layer(_Holder):
    def __init__(self, D: int, F: int):
        super().__init__()
        self.attention = _Attention(D)
        self.layernorm_before = LayerNormParams(D)
        self.layernorm_after = LayerNormParams(D)
        self.mlp = _MLP(D, F)


class ViTParams(_Holder):
    """Same parameter names as HF ``ViTModel`` (pooler removed: the reference sets it to Identity, backbones.py:283)."""

    def __init__(self, D: int, heads: int, F: int, has_cls: bool = True, num_reg_tokens: int = 0):
        super().__init__()
        self.hidden_size, self.num_heads, self.intermediate_size = D, heads, F
        self.has_cls, self.num_reg_tokens = has_cls, num_reg_tokens
        self.embeddings = _Embeddings(D, has_cls, num_reg_tokens)
        self.layers = nn.ModuleList([_Layer(D, F) for _ in range(NUM_LAYERS)])
        self.layernorm = LayerNormParams(D)
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """HF ViT ``_init_weights``: trunc-normal(std 0.02) matrices / pos-emb / cls, zero biases, unit LayerNorm."""
        for name, p in self.named_parameters():
            if "layernorm" in name:
                (nn.init.ones_ if name.endswith("weight") else nn.init.zeros_)(p)
            elif name.endswith("bias"):
                nn.init.zeros_(p)
            else:
                nn.init.trunc_normal_(p, mean=0.0, std=INIT_RANGE)


# reference-era (transformers 4.4x) checkpoint key -> current key  (not verifiable offline)
def remap_legacy_key(k: str) -> str:
    k = k.replace(".encoder.layer.", ".layers.")
    k = k.replace(".attention.attention.query.", ".attention.q_proj.")
    k = k.replace(".attention.attention.key.", ".attention.k_proj.")
    k = k.replace(".attention.attention.value.", ".attention.v_proj.")
    k = k.replace(".attention.output.dense.", ".attention.o_proj.")
    k = k.replace(".intermediate.dense.", ".mlp.fc1.")
    if ".layers." in k and ".output.dense." in k:
        k = k.replace(".output.dense.", ".mlp.fc2.")
    return k


DEIT_HUB_PROCESSOR = {"size": 256, "crop_size": 224, "resample": 3,
                      "image_mean": (0.485, 0.456, 0.406), "image_std": (0.229, 0.224, 0.225)}


class DeiT(nn.Module):
    """DeiT student (reference ``DeiT`` backbones.py:255-341).  ``forward`` takes uint8 images and returns the last
    hidden state [B, 197, D]; image preprocessing (rescale + normalise) is fused into the ingest kernel.

    The ``nocls-`` (``DeiTNoCLS``, backbones.py:344-416: no CLS token, 196 tokens) and ``reg-`` (``DeiTReg``, :419-503: 7
    register tokens appended after the patches, 204 tokens) students are the same container with a different token layout;
    they carry the reference's marker attributes (``no_cls`` / ``num_reg_tokens``) that ``RobotVisionFM`` keys on."""

    def __init__(self, model_name: str = "facebook/deit-small-patch16-224", pretrained: bool = False, image_size: int = 224,
                 processor: Optional[dict] = None, num_reg_tokens: int = 7):
        super().__init__()
        base, has_cls, nreg = backbone_variant(model_name)
        nreg = int(num_reg_tokens) if nreg is None else nreg
        if base not in ARCH:
            raise NotImplementedError(f"Requested {model_name} is not implemented.")
        if pretrained:
            raise NotImplementedError("pretrained HF weights cannot be fetched offline; load a checkpoint with "
                                      "RobotVisionFM.load_pretrained_weights instead")
        self.model_name = model_name
        self.image_size = image_size
        D, heads, F = ARCH[base]
        self.model = ViTParams(D, heads, F, has_cls, nreg)
        if not has_cls:
            self.no_cls = True  # reference marker attribute (RobotVisionFM: hasattr(backbone, "no_cls"))
        if nreg > 0:
            self.num_reg_tokens = nreg
        self._engine = None  # set by RobotVisionFM
        # The reference takes its image processor from the hub (AutoProcessor.from_pretrained, backbones.py:292); nothing can
        # be fetched here, so the configuration is explicit.  Default: resize to 224x224 bilinear, no
        # crop, ImageNet mean/std (the ViT processor the golden vectors were generated with).  processor="deit" (or a dict)
        # selects the preprocessor_config.json of the facebook/deit-*-patch16-224 checkpoints: resize 256 bicubic +
        # center-crop 224 (transformers DeiTImageProcessor; golden G14).
        self.set_processor(**({} if processor is None else DEIT_HUB_PROCESSOR if processor == "deit" else dict(processor)))

    def set_processor(self, size=224, crop_size: Optional[int] = None, resample: int = 2,
                      image_mean=(0.485, 0.456, 0.406), image_std=(0.229, 0.224, 0.225)) -> None:
        """size: int or (h, w) the resize target; crop_size: center-crop edge or None; resample: 2 bilinear / 3 bicubic."""
        self.resize_size = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
        self.crop_size = None if crop_size is None else int(crop_size)
        if resample not in (2, 3):
            raise NotImplementedError(f"resample={resample}: PIL BILINEAR (2) and BICUBIC (3) are implemented")
        self.resample = int(resample)
        final = (self.crop_size, self.crop_size) if self.crop_size else self.resize_size
        if final != (224, 224):
            raise NotImplementedError(f"processor output {final}: the student consumes 224x224 (no pos-emb interpolation)")
        self.image_mean, self.image_std = tuple(float(v) for v in image_mean), tuple(float(v) for v in image_std)
        if self._engine is not None:
            self._engine._luts.clear()

    def get_feature_size(self, keep_spatial: bool = False, return_torch_size: bool = False):
        D = self.model.hidden_size
        size = (D, GRID * GRID)
        if keep_spatial:
            assert math.isqrt(size[-1])
            size = (D, GRID, GRID)
            if return_torch_size:
                size = torch.Size(size)
        return size

    def forward(self, x: Any, do_resize: bool = True, interpolate_pos_encoding: Optional[bool] = None, do_rescale: bool = True,
                do_normalize: bool = True) -> torch.Tensor:
        """-> last hidden state [B, tokens, D].  ``interpolate_pos_encoding=True`` (with ``do_resize=False``) runs the student
        on images of another size: the 14x14 patch position table is interpolated bicubically to the image's patch grid
        (HF ViTEmbeddings.interpolate_pos_encoding for DeiT; the reference's own variant for nocls- / reg-)."""
        if self._engine is None:
            raise RuntimeError("DeiT is driven by RobotVisionFM's engine; construct it through RobotVisionFM")
        return self._engine.backbone(x, do_rescale=do_rescale, do_normalize=do_normalize, do_resize=do_resize,
                                     interpolate_pos_encoding=bool(interpolate_pos_encoding))


DeiTNoCLS = DeiT  # same container; the model name selects the token layout
DeiTReg = DeiT


def build_backbone(model_name: str, pretrained: bool = False, image_size: int = 224, **kwargs: Any) -> nn.Module:
    """Reference ``build_backbone`` backbones.py:506-526: "reg" names -> DeiTReg (kwarg num_reg_tokens), "nocls" -> DeiTNoCLS,
    "deit" -> DeiT."""
    if "reg" in model_name or "nocls" in model_name or "deit" in model_name:
        extra = {"num_reg_tokens": kwargs["num_reg_tokens"]} if "num_reg_tokens" in kwargs and "reg" in model_name else {}
        return DeiT(model_name=model_name, pretrained=pretrained, image_size=image_size, processor=kwargs.get("processor"), **extra)
    raise NotImplementedError(f"Requested {model_name} is not implemented.")
