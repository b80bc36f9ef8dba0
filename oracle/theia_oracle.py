"""CPU ORACLE for the Theia distillation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain PyTorch fp32 tensor ops, NHWC /
token-major layout, fused QKV, convolutions written out as shifted matmuls) of the ONE
hot path this repository accelerates:

    uint8 image -> preprocess -> DeiT/ViT student -> per-teacher ``lconv`` translator heads
    -> {mse, smooth-L1, cosine} feature-matching losses -> backward.

It is the *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package ``theia_amd`` never does and fails
loudly when its HIP extension is missing.

Parity status: **pinned by generated golden vectors.**  The reference repository has no
tests / known-answer vectors of its own, and its arithmetic for the
backbone lives in the third-party ``transformers`` package (unpinned in the reference's
pyproject.toml:18; 5.15.0 installed here) plus ``torch.nn``.  This oracle is therefore pinned
against outputs of the reference itself, imported in the build container by
``oracle/gen_golden.py`` (which writes ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
re-checks the oracle against those fixtures everywhere, including the GPU box.

Reference call sites each function follows (paths relative to /root/reference/src/theia):
    preprocess              models/backbones.py:337-339  (+ HF image_transforms rescale/normalize)
    vit_forward             models/backbones.py:340-341  (+ HF modeling_vit ViTEmbeddings/ViTLayer)
    handle_feature_output   models/utils.py:8-43
    head_forward            models/adapter_heads.py:232-359 (LightConvAdapterHead)
    translator_forward      models/feature_translators.py:68-88,159-205
    get_loss                models/rvfm.py:138-185
    main_loss               scripts/train/train_rvfm.py:119-122
    MODEL_FEATURE_SIZES     foundation_models/common.py:18-25
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------------------
# constants (foundation_models/common.py:18-25 ; configs/training/target_models/*.yaml)
# ----------------------------------------------------------------------------------------
MODEL_FEATURE_SIZES: Dict[str, Tuple[int, int, int]] = {
    "facebook/dinov2-large": (1024, 16, 16),
    "facebook/sam-vit-huge": (256, 64, 64),
    "google/vit-huge-patch14-224-in21k": (1280, 16, 16),
    "llava-hf/llava-1.5-7b-hf": (1024, 24, 24),
    "openai/clip-vit-large-patch14": (1024, 16, 16),
    "LiheYoung/depth-anything-large-hf": (32, 64, 64),
}

TEACHER_SETS: Dict[str, List[str]] = {
    # configs/training/target_models/dinov2.yaml:1-3
    "dinov2": ["facebook/dinov2-large"],
    # configs/training/target_models/cdiv.yaml:1-5
    "cdiv": [
        "google/vit-huge-patch14-224-in21k",
        "facebook/dinov2-large",
        "openai/clip-vit-large-patch14",
    ],
    # configs/training/target_models/cddsv.yaml:1-7
    "cddsv": [
        "google/vit-huge-patch14-224-in21k",
        "facebook/dinov2-large",
        "openai/clip-vit-large-patch14",
        "facebook/sam-vit-huge",
        "LiheYoung/depth-anything-large-hf",
    ],
}

# student sizes: (hidden D, heads, mlp F).  HF ViTConfig defaults for everything else
# (12 layers, patch 16, image 224, eps 1e-12, erf GELU, qkv_bias) -- SURVEY.md §8(c).
ARCH: Dict[str, Tuple[int, int, int]] = {
    "facebook/deit-tiny-patch16-224": (192, 3, 768),
    "facebook/deit-small-patch16-224": (384, 6, 1536),
    "facebook/deit-base-patch16-224": (768, 12, 3072),
}
NUM_REG_TOKENS = 7  # DeiTReg default (backbones.py:419)


def backbone_variant(backbone: str) -> "Tuple[str, bool, int]":
    """(base model name, has CLS token, number of register tokens) of a backbone name: "nocls-<name>" is the student without
    CLS token (backbones.py:344-372), "reg-<name>" the one with 7 register tokens appended (backbones.py:419-451);
    dispatch order as build_backbone (:519-526): "reg" is tested first."""
    if backbone.startswith("reg-"):
        return backbone[4:], True, NUM_REG_TOKENS
    if backbone.startswith("nocls-"):
        return backbone[6:], False, 0
    return backbone, True, 0


def _arch(backbone: str) -> "Tuple[int, int, int]":
    return ARCH[backbone_variant(backbone)[0]]


NUM_LAYERS = 12
PATCH = 16
IMAGE = 224
GRID = IMAGE // PATCH  # 14
NTOK = 1 + GRID * GRID  # 197
LN_EPS_VIT = 1e-12
LN_EPS_HEAD = 1e-5
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def head_key(teacher: str) -> str:
    """feature_translators.py:46 -- module name = teacher name with '.' -> '_'."""
    return teacher.replace(".", "_")


# ----------------------------------------------------------------------------------------
# deterministic, platform-independent synthetic parameters (integer hash -> uniform)
# ----------------------------------------------------------------------------------------
def _hash_uniform(n: int, seed: int) -> np.ndarray:
    """n float32 values in [-1, 1) from a counter-based 64-bit mix (splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1) with 24 bits
    return (u * 2.0 - 1.0).astype(np.float32)


def param_shapes(backbone: str, teachers: Sequence[str]) -> "Dict[str, Tuple[int, ...]]":
    """Canonical (reference state_dict) parameter names and shapes, in registration order.

    Names are those observed for the reference under transformers 5.x.
    """
    D, _h, F = _arch(backbone)
    _base, has_cls, nreg = backbone_variant(backbone)
    shapes: Dict[str, Tuple[int, ...]] = {}
    e = "backbone.model.embeddings."
    if has_cls:  # ViTEmbeddingsNoCLS sets cls_token = None (backbones.py:36): the parameter disappears from the state_dict
        shapes[e + "cls_token"] = (1, 1, D)
    shapes[e + "position_embeddings"] = (1, NTOK, D)
    if nreg:  # ViTEmbeddingsReg (backbones.py:134-137), registered after position_embeddings
        shapes[e + "reg_token"] = (1, nreg, D)
        shapes[e + "reg_pos_embed"] = (1, nreg, D)
    shapes[e + "patch_embeddings.projection.weight"] = (D, 3, PATCH, PATCH)
    shapes[e + "patch_embeddings.projection.bias"] = (D,)
    for i in range(NUM_LAYERS):
        p = f"backbone.model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            shapes[p + f"attention.{n}.weight"] = (D, D)
            shapes[p + f"attention.{n}.bias"] = (D,)
        shapes[p + "layernorm_before.weight"] = (D,)
        shapes[p + "layernorm_before.bias"] = (D,)
        shapes[p + "layernorm_after.weight"] = (D,)
        shapes[p + "layernorm_after.bias"] = (D,)
        shapes[p + "mlp.fc1.weight"] = (F, D)
        shapes[p + "mlp.fc1.bias"] = (F,)
        shapes[p + "mlp.fc2.weight"] = (D, F)
        shapes[p + "mlp.fc2.bias"] = (D,)
    shapes["backbone.model.layernorm.weight"] = (D,)
    shapes["backbone.model.layernorm.bias"] = (D,)
    C = D  # hidden_size_factor = 1.0 (configs/model/translator/lconv.yaml)
    for t in teachers:
        p = f"translator.translator_heads.{head_key(t)}."
        if t.endswith("_cls"):  # LinearAdapterHead on token 0 (adapter_heads.py:28-58; names from train_rvfm.py:238-246)
            Ct = MODEL_FEATURE_SIZES[t[:-4]][0]
            shapes[p + "adapter.0.weight"] = (Ct, C)
            shapes[p + "adapter.0.bias"] = (Ct,)
            continue
        Ct, Ht, _Wt = MODEL_FEATURE_SIZES[t]
        shapes[p + "pad.1.weight"] = (C, C, 3, 3)
        shapes[p + "pad.1.bias"] = (C,)
        if Ht == 16:
            sizes = (16, 16, 16)
        elif Ht == 64:
            sizes = (16, 31, 64)
        else:
            raise NotImplementedError(f"teacher spatial size {Ht}")
        shapes[p + "adapter.0.weight"] = (C, sizes[0], sizes[0])
        shapes[p + "adapter.0.bias"] = (C, sizes[0], sizes[0])
        shapes[p + "adapter.1.weight"] = (C, C, 3, 3)
        shapes[p + "adapter.1.bias"] = (C,)
        shapes[p + "adapter.3.weight"] = (C, sizes[1], sizes[1])
        shapes[p + "adapter.3.bias"] = (C, sizes[1], sizes[1])
        shapes[p + "adapter.4.weight"] = (C, C, 3, 3)
        shapes[p + "adapter.4.bias"] = (C,)
        shapes[p + "adapter.6.weight"] = (C, sizes[2], sizes[2])
        shapes[p + "adapter.6.bias"] = (C, sizes[2], sizes[2])
        shapes[p + "adapter.8.weight"] = (Ct, C)
        shapes[p + "adapter.8.bias"] = (Ct,)
    return shapes


def synth_params(backbone: str, teachers: Sequence[str], seed: int = 0) -> "Dict[str, torch.Tensor]":
    """Deterministic synthetic weights, reproducible without torch RNG.

    Scales are chosen so activations stay O(1) through 12 layers and the heads (matrices:
    U(-a,a) with a = sqrt(3/fan_in); biases / pos-emb / cls: small; LN weights 1 +- 0.1)."""
    out: Dict[str, torch.Tensor] = {}
    for idx, (name, shape) in enumerate(param_shapes(backbone, teachers).items()):
        n = int(np.prod(shape))
        u = _hash_uniform(n, seed * 100003 + idx + 1)
        leaf = name.rsplit(".", 1)[-1]
        is_ln = ("layernorm" in name) or (any(f"adapter.{k}." in name for k in (0, 3, 6)) and len(shape) == 3)
        if is_ln and leaf == "weight":
            v = 1.0 + 0.1 * u
        elif is_ln and leaf == "bias":
            v = 0.05 * u
        elif leaf == "bias":
            v = 0.02 * u
        elif leaf in ("cls_token", "position_embeddings", "reg_token", "reg_pos_embed"):
            v = 0.5 * u
        else:
            if len(shape) == 4 and "patch_embeddings" in name:
                fan_in = shape[1] * shape[2] * shape[3]
            elif len(shape) == 4 and ("pad.1" in name or _is_convT(name, shape, teachers)):
                fan_in = shape[0] * 9 / 2.0  # transposed conv: [Cin, Cout, 3, 3]
            elif len(shape) == 4:
                fan_in = shape[1] * 9
            else:
                fan_in = shape[1]
            v = math.sqrt(3.0 / fan_in) * u
        out[name] = torch.from_numpy(np.ascontiguousarray(v.reshape(shape).astype(np.float32)))
    return out


def _is_convT(name: str, shape, teachers) -> bool:
    for t in teachers:
        if not t.endswith("_cls") and head_key(t) + "." in name and MODEL_FEATURE_SIZES[t][1] == 64:
            return True
    return False


def synth_images(b: int, seed: int = 0) -> torch.Tensor:
    """uint8 [b,224,224,3] from the integer hash (not torch RNG)."""
    u = _hash_uniform(b * IMAGE * IMAGE * 3, 7919 + seed)
    v = np.floor((u.astype(np.float64) + 1.0) * 128.0).clip(0, 255).astype(np.uint8)
    return torch.from_numpy(v.reshape(b, IMAGE, IMAGE, 3))


def synth_targets(b: int, teachers: Sequence[str], seed: int = 1) -> "Dict[str, torch.Tensor]":
    """fp32 teacher features [b, H*W, Ct], roughly unit variance, from the integer hash."""
    out = {}
    for i, t in enumerate(teachers):
        if t.endswith("_cls"):  # the teacher's CLS token: [b, Ct]
            Ct = MODEL_FEATURE_SIZES[t[:-4]][0]
            u = _hash_uniform(b * Ct, 104729 * (seed + 1) + i)
            out[t] = torch.from_numpy((u * math.sqrt(3.0)).reshape(b, Ct).astype(np.float32))
            continue
        Ct, Ht, Wt = MODEL_FEATURE_SIZES[t]
        n = b * Ht * Wt * Ct
        u = _hash_uniform(n, 104729 * (seed + 1) + i)
        out[t] = torch.from_numpy((u * math.sqrt(3.0)).reshape(b, Ht * Wt, Ct).astype(np.float32))
    return out


# ----------------------------------------------------------------------------------------
# a1: preprocessing (HF PIL backend at 224x224: rescale -> normalize; resize is identity)
# ----------------------------------------------------------------------------------------
def preprocess_lut(do_rescale: bool = True, do_normalize: bool = True,
                   mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD) -> np.ndarray:
    """[3,256] float32 table: value of channel c for uint8 input v.

    Follows transformers/image_transforms.py rescale (:118-122: float64 multiply then cast to
    float32) and normalize (:419-439: float32 (x-mean)/std with float32 mean/std)."""
    v = np.arange(256, dtype=np.uint8)
    lut = np.zeros((3, 256), dtype=np.float32)
    for c in range(3):
        x = v
        if do_rescale:
            x = (x.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
        if do_normalize:
            if not np.issubdtype(x.dtype, np.floating):
                x = x.astype(np.float32)
            m = np.array(mean[c], dtype=x.dtype)
            s = np.array(std[c], dtype=x.dtype)
            x = (x - m) / s
        lut[c] = x.astype(np.float32)
    return lut


def to_bhwc_uint8(x) -> torch.Tensor:
    """Accept what DeiT.forward accepts (backbones.py:314-341): uint8 torch [B,H,W,C] or [B,C,H,W],
    a single [H,W,C]/[C,H,W] image, numpy arrays, or a list of PIL images / arrays.
    Channel layout inference mirrors transformers/image_utils.py:288-324 (first dim in (1,3) ->
    channels-first, else last)."""
    if isinstance(x, (list, tuple)):
        imgs = [to_bhwc_uint8(i)[0] for i in x]
        return torch.stack(imgs, 0)
    if not isinstance(x, (torch.Tensor, np.ndarray)):
        x = np.asarray(x)  # PIL image
        if x.ndim == 2:
            x = np.stack([x] * 3, -1)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.dim() == 3:
        x = x.unsqueeze(0)
    assert x.dim() == 4, f"expected a 3-D or 4-D image tensor, got {tuple(x.shape)}"
    if x.shape[1] in (1, 3) and x.shape[-1] not in (1, 3):
        x = x.permute(0, 2, 3, 1)
    elif x.shape[1] in (1, 3) and x.shape[-1] in (1, 3):
        # ambiguous: HF checks the first candidate axis first (channels-first wins)
        x = x.permute(0, 2, 3, 1)
    assert x.dtype == torch.uint8, "oracle handles uint8 images only"
    return x.contiguous()


def preprocess(images, do_rescale: bool = True, do_normalize: bool = True, any_size: bool = False) -> torch.Tensor:
    """uint8 images -> fp32 [b,H,W,3] (NHWC); H = W = 224 unless any_size (do_resize=False + interpolate_pos_encoding)."""
    x = to_bhwc_uint8(images)
    assert any_size or (x.shape[1] == IMAGE and x.shape[2] == IMAGE), "oracle covers 224x224 inputs (resize is identity)"
    lut = torch.from_numpy(preprocess_lut(do_rescale, do_normalize))  # [3,256]
    idx = x.long()
    out = torch.stack([lut[c][idx[..., c]] for c in range(3)], dim=-1)
    return out


# ----------------------------------------------------------------------------------------
# a2-a4: ViT student
# ----------------------------------------------------------------------------------------
def _layernorm_rows(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def patch_matrix(pix: torch.Tensor) -> torch.Tensor:
    """[b,H,W,3] fp32 -> [b,(H/16)*(W/16),768]; K index = c*256 + ky*16 + kx; token p = py*(W/16)+px."""
    b, H, W = pix.shape[0], pix.shape[1], pix.shape[2]
    gh, gw = H // PATCH, W // PATCH
    x = pix.view(b, gh, PATCH, gw, PATCH, 3)  # b py ky px kx c
    x = x.permute(0, 1, 3, 5, 2, 4)  # b py px c ky kx
    return x.reshape(b, gh * gw, 3 * PATCH * PATCH)


def interpolate_patch_pos(pos_patches: torch.Tensor, gh: int, gw: int, flavour: str) -> torch.Tensor:
    """Bicubic interpolation of the 14x14 patch position table [196, D] to gh x gw -> [gh*gw, D] (align_corners=False).
    flavour "size": HF ViTEmbeddings.interpolate_pos_encoding (transformers modeling_vit.py:89-127, what the reference's DeiT
    runs): F.interpolate(size=(gh, gw)).  flavour "scale": the reference's own NoCLS / Reg embeddings (backbones.py:39-69,
    146-177): scale_factor = ((gh + 0.1) / 14, (gw + 0.1) / 14), which samples at slightly different coordinates."""
    D = pos_patches.shape[-1]
    t = pos_patches.reshape(1, GRID, GRID, D).permute(0, 3, 1, 2)
    if flavour == "size":
        t = torch.nn.functional.interpolate(t, size=(gh, gw), mode="bicubic", align_corners=False)
    else:
        t = torch.nn.functional.interpolate(t, scale_factor=((gh + 0.1) / GRID, (gw + 0.1) / GRID), mode="bicubic", align_corners=False)
        assert t.shape[-2] == gh and t.shape[-1] == gw
    return t.permute(0, 2, 3, 1).reshape(gh * gw, D)


def vit_forward(params: Dict[str, torch.Tensor], pix: torch.Tensor, backbone: str, interpolate_pos_encoding: bool = False) -> torch.Tensor:
    """fp32 [b,H,W,3] -> last_hidden_state [b, ntok, D] (modeling_vit.py ViTModel, pooler=Identity).  Token layout:
    [CLS (unless nocls-)] + patches + [7 register tokens (reg-)]  (backbones.py:71-95, 179-209)."""
    D, nh, F = _arch(backbone)
    _base, has_cls, nreg = backbone_variant(backbone)
    dh = D // nh
    b = pix.shape[0]
    gh, gw = pix.shape[1] // PATCH, pix.shape[2] // PATCH
    e = "backbone.model.embeddings."
    Wp = params[e + "patch_embeddings.projection.weight"].reshape(D, -1)
    emb = patch_matrix(pix) @ Wp.t() + params[e + "patch_embeddings.projection.bias"]
    pos = params[e + "position_embeddings"][0]  # [197, D]
    if interpolate_pos_encoding and not (gh * gw == GRID * GRID and gh == gw):
        ppos = interpolate_patch_pos(pos[1:], gh, gw, "size" if (has_cls and nreg == 0) else "scale")
    else:
        assert gh == GRID and gw == GRID, "non-224 input needs interpolate_pos_encoding"
        ppos = pos[1:]
    toks = [emb + ppos]
    if has_cls:
        toks.insert(0, (params[e + "cls_token"][0] + pos[:1]).expand(b, 1, D))
    if nreg:
        toks.append((params[e + "reg_token"][0] + params[e + "reg_pos_embed"][0]).expand(b, nreg, D))
    h = torch.cat(toks, dim=1)
    NTOK = h.shape[1]  # noqa: N806 (shadows the 224-input constant inside this function)
    for i in range(NUM_LAYERS):
        p = f"backbone.model.layers.{i}."
        a = _layernorm_rows(h, params[p + "layernorm_before.weight"], params[p + "layernorm_before.bias"], LN_EPS_VIT)
        Wqkv = torch.cat([params[p + f"attention.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)
        bqkv = torch.cat([params[p + f"attention.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0)
        qkv = a @ Wqkv.t() + bqkv  # [b,197,3D]
        q, k, v = qkv.split(D, dim=-1)
        q = q.view(b, NTOK, nh, dh).transpose(1, 2)
        k = k.view(b, NTOK, nh, dh).transpose(1, 2)
        v = v.view(b, NTOK, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        pr = torch.softmax(s, dim=-1)
        o = (pr @ v).transpose(1, 2).reshape(b, NTOK, D)
        h = h + (o @ params[p + "attention.o_proj.weight"].t() + params[p + "attention.o_proj.bias"])
        m = _layernorm_rows(h, params[p + "layernorm_after.weight"], params[p + "layernorm_after.bias"], LN_EPS_VIT)
        m = _gelu_erf(m @ params[p + "mlp.fc1.weight"].t() + params[p + "mlp.fc1.bias"])
        h = h + (m @ params[p + "mlp.fc2.weight"].t() + params[p + "mlp.fc2.bias"])
    return _layernorm_rows(h, params["backbone.model.layernorm.weight"], params["backbone.model.layernorm.bias"], LN_EPS_VIT)


# ----------------------------------------------------------------------------------------
# a5: token selection (models/utils.py:8-43)
# ----------------------------------------------------------------------------------------
def handle_feature_output(x: torch.Tensor, feature_reduce_method: Optional[str] = None,
                          num_discard_tokens: int = 0) -> torch.Tensor:
    n = x.shape[1]
    if feature_reduce_method == "mean_pooling":
        return x[:, 1:n - num_discard_tokens].mean(dim=1)
    if feature_reduce_method == "max_pooling":
        return x[:, 1:n - num_discard_tokens].amax(dim=1)
    if feature_reduce_method == "cls":
        return x[:, 0]
    if feature_reduce_method == "identity":
        return x
    if feature_reduce_method is None:
        return x[:, 1:n - num_discard_tokens]
    raise NotImplementedError(f"feature_reduce_method {feature_reduce_method} it not implemented.")


# ----------------------------------------------------------------------------------------
# a9: LightConvAdapterHead, NHWC, convolutions as shifted matmuls (adapter_heads.py:232-359)
# ----------------------------------------------------------------------------------------
def _shift_gather(x: torch.Tensor, OH: int, OW: int, sy: int, sx: int, dy: int, dx: int) -> torch.Tensor:
    """x [b,IH,IW,C] -> g [b,OH,OW,C] with g[b,oy,ox] = x[b, oy*sy+dy, ox*sx+dx] (0 outside)."""
    b, IH, IW, C = x.shape
    oy = torch.arange(OH) * sy + dy
    ox = torch.arange(OW) * sx + dx
    vy = (oy >= 0) & (oy < IH)
    vx = (ox >= 0) & (ox < IW)
    g = x[:, oy.clamp(0, IH - 1)][:, :, ox.clamp(0, IW - 1)]
    mask = (vy[:, None] & vx[None, :]).to(x.dtype)[None, :, :, None]
    return g * mask


def conv3x3_p1(x: torch.Tensor, W: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(C,C,3,padding=1), W [co,ci,ky,kx]; x NHWC."""
    b, H, Wd, C = x.shape
    out = bias.view(1, 1, 1, -1).expand(b, H, Wd, W.shape[0]).clone()
    for ky in range(3):
        for kx in range(3):
            out = out + _shift_gather(x, H, Wd, 1, 1, ky - 1, kx - 1) @ W[:, :, ky, kx].t()
    return out


def convT3x3(x: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, stride: int, padding: int,
             output_padding: int) -> torch.Tensor:
    """nn.ConvTranspose2d(C,C,3,stride,padding,output_padding), W [ci,co,ky,kx]; x NHWC.

    out[oy,ox,co] = bias + sum_{ky,kx,ci} x[i,j,ci] W[ci,co,ky,kx] with oy = i*stride - padding + ky.
    Written in gather form per output-parity class (stride 2) / directly (stride 1)."""
    b, IH, IW, C = x.shape
    OH = (IH - 1) * stride - 2 * padding + 3 + output_padding
    OW = (IW - 1) * stride - 2 * padding + 3 + output_padding
    Co = W.shape[1]
    out = bias.view(1, 1, 1, -1).expand(b, OH, OW, Co).clone()
    if stride == 1:
        for ky in range(3):
            for kx in range(3):
                out = out + _shift_gather(x, OH, OW, 1, 1, padding - ky, padding - kx) @ W[:, :, ky, kx]
        return out
    assert stride == 2
    for py in range(2):
        for px in range(2):
            nyc = len(range(py, OH, 2))
            nxc = len(range(px, OW, 2))
            acc = torch.zeros(b, nyc, nxc, Co, dtype=x.dtype)
            for ky in range(3):
                if (py + padding - ky) % 2 != 0:
                    continue
                dyc = (py + padding - ky) // 2
                for kx in range(3):
                    if (px + padding - kx) % 2 != 0:
                        continue
                    dxc = (px + padding - kx) // 2
                    acc = acc + _shift_gather(x, nyc, nxc, 1, 1, dyc, dxc) @ W[:, :, ky, kx]
            out[:, py::2, px::2, :] = out[:, py::2, px::2, :] + acc
    return out


def layernorm_chw(x: torch.Tensor, w_chw: torch.Tensor, b_chw: torch.Tensor, eps: float = LN_EPS_HEAD) -> torch.Tensor:
    """nn.LayerNorm([C,H,W]) on an NHWC activation; affine params stay in reference [C,H,W] layout."""
    b = x.shape[0]
    flat = x.reshape(b, -1)
    mu = flat.mean(dim=1).view(b, 1, 1, 1)
    var = ((flat - flat.mean(dim=1, keepdim=True)) ** 2).mean(dim=1).view(b, 1, 1, 1)
    xh = (x - mu) / torch.sqrt(var + eps)
    return xh * w_chw.permute(1, 2, 0) + b_chw.permute(1, 2, 0)


def head_forward(params: Dict[str, torch.Tensor], z: torch.Tensor, teacher: str, backbone_no_cls: bool = False) -> torch.Tensor:
    """z [b,197,C] (final LN output incl. CLS) -> predicted teacher feature [b, Ht*Wt, Ct] (or [b, Ct] for a "_cls" head:
    one Linear on token 0, adapter_heads.py:50-57)."""
    p = f"translator.translator_heads.{head_key(teacher)}."
    if teacher.endswith("_cls"):
        return z[:, 0] @ params[p + "adapter.0.weight"].t() + params[p + "adapter.0.bias"]
    Ct, Ht, Wt = MODEL_FEATURE_SIZES[teacher]
    b, _n, C = z.shape
    x = (z if backbone_no_cls else z[:, 1:, :]).reshape(b, GRID, GRID, C)  # adapter_heads.py:355-356 + Rearrange :281
    # pad: ConvTranspose2d(C,C,3,stride=1,output_padding=0): 14 -> 16 (adapter_heads.py:279-290)
    x = convT3x3(x, params[p + "pad.1.weight"], params[p + "pad.1.bias"], 1, 0, 0)
    x = layernorm_chw(x, params[p + "adapter.0.weight"], params[p + "adapter.0.bias"])
    if Ht == 64:  # adapter_heads.py:304-315
        x = torch.relu(convT3x3(x, params[p + "adapter.1.weight"], params[p + "adapter.1.bias"], 2, 1, 0))
        x = layernorm_chw(x, params[p + "adapter.3.weight"], params[p + "adapter.3.bias"])
        x = torch.relu(convT3x3(x, params[p + "adapter.4.weight"], params[p + "adapter.4.bias"], 2, 0, 1))
        x = layernorm_chw(x, params[p + "adapter.6.weight"], params[p + "adapter.6.bias"])
    elif Ht == 16:  # adapter_heads.py:316-327
        x = torch.relu(conv3x3_p1(x, params[p + "adapter.1.weight"], params[p + "adapter.1.bias"]))
        x = layernorm_chw(x, params[p + "adapter.3.weight"], params[p + "adapter.3.bias"])
        x = torch.relu(conv3x3_p1(x, params[p + "adapter.4.weight"], params[p + "adapter.4.bias"]))
        x = layernorm_chw(x, params[p + "adapter.6.weight"], params[p + "adapter.6.bias"])
    else:
        raise NotImplementedError
    x = x.reshape(b, Ht * Wt, C)
    return x @ params[p + "adapter.8.weight"].t() + params[p + "adapter.8.bias"]


def translator_forward(params, z, teachers: Sequence[str], backbone_no_cls: bool = False) -> "Dict[str, torch.Tensor]":
    return {t: head_forward(params, z, t, backbone_no_cls) for t in teachers}


# ----------------------------------------------------------------------------------------
# a6, a7: model entry points (models/rvfm.py:94-136)
# ----------------------------------------------------------------------------------------
def forward_feature(params, images, backbone: str, feature_reduce_method: Optional[str] = None,
                    do_rescale: bool = True, do_normalize: bool = True, interpolate_pos_encoding: bool = False) -> torch.Tensor:
    """models/rvfm.py:94-113.  Note the reference slices x[:, 1:n-disc] whatever the backbone: a nocls- student loses its
    first patch token here (195 tokens) -- reproduced, not fixed."""
    z = vit_forward(params, preprocess(images, do_rescale, do_normalize, any_size=interpolate_pos_encoding), backbone,
                    interpolate_pos_encoding)
    return handle_feature_output(z, feature_reduce_method, backbone_variant(backbone)[2])


def forward(params, images, backbone: str, teachers: Sequence[str]) -> "Dict[str, torch.Tensor]":
    """models/rvfm.py:115-136: register tokens are stripped before the translator (:133-134)."""
    z = vit_forward(params, preprocess(images), backbone)
    _base, has_cls, nreg = backbone_variant(backbone)
    if nreg:
        z = z[:, :-nreg]
    return translator_forward(params, z, teachers, backbone_no_cls=not has_cls)


# ----------------------------------------------------------------------------------------
# a10, a11: losses (models/rvfm.py:138-185 ; train_rvfm.py:119-122)
# ----------------------------------------------------------------------------------------
def _smooth_l1(d: torch.Tensor) -> torch.Tensor:
    a = d.abs()
    return torch.where(a < 1.0, 0.5 * d * d, a - 0.5)


def get_loss(pred: Dict[str, torch.Tensor], y: Dict[str, torch.Tensor]) -> dict:
    T = len(pred)
    mse_avg = 0.0
    cos_avg = 0.0
    l1_avg = 0.0
    mse_pm, cos_pm, l1_pm = {}, {}, {}
    for t in pred:
        p, q = pred[t], y[t]
        d = p - q
        mse = (d * d).mean()  # nn.MSELoss (rvfm.py:158)
        l1 = _smooth_l1(d).mean()  # nn.SmoothL1Loss beta=1 (rvfm.py:162)
        pf = p.flatten(1)
        qf = q.flatten(1)
        # F.normalize(eps=1e-12) (rvfm.py:165-166)
        pn = pf / pf.norm(dim=1, keepdim=True).clamp_min(1e-12)
        qn = qf / qf.norm(dim=1, keepdim=True).clamp_min(1e-12)
        # nn.CosineEmbeddingLoss, target=+1, its own EPSILON=1e-12 on squared norms (rvfm.py:167-168)
        dot = (pn * qn).sum(1)
        m1 = (pn * pn).sum(1) + 1e-12
        m2 = (qn * qn).sum(1) + 1e-12
        cos = (1.0 - dot / torch.sqrt(m1 * m2)).mean()
        w = 1.0 / T
        mse_avg = mse_avg + mse * w
        cos_avg = cos_avg + cos / T
        l1_avg = l1_avg + l1 * w
        mse_pm[t], cos_pm[t], l1_pm[t] = float(mse), float(cos), float(l1)
    return {
        "mse_loss": mse_avg, "cos_loss": cos_avg, "l1_loss": l1_avg,
        "mse_losses_per_model": mse_pm, "cos_losses_per_model": cos_pm, "l1_losses_per_model": l1_pm,
    }


def main_loss(losses: dict, kind: Optional[str] = "cos_l1") -> torch.Tensor:
    if kind == "mse" or kind is None:
        return losses["mse_loss"]
    if kind == "cos_l1":
        return 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
    raise ValueError(kind)


def train_step_grads(params: Dict[str, torch.Tensor], images, targets: Dict[str, torch.Tensor], backbone: str,
                     teachers: Sequence[str], loss_kind: str = "cos_l1"):
    """fwd + loss + bwd (train_rvfm.py:116-125). Returns (losses, main, grads-by-name, pred)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pred = forward(leaf, images, backbone, teachers)
    losses = get_loss(pred, targets)
    main = main_loss(losses, loss_kind)
    main.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return losses, main.detach(), grads, {k: v.detach() for k, v in pred.items()}


# ----------------------------------------------------------------------------------------
# K14 / §8(f)-2: bf16 teacher-feature normalisation (dataset/data_utils.py:342-355,374-379)
# ----------------------------------------------------------------------------------------
def ingest_feature_chw_bf16(x_chw_bf16: torch.Tensor, mean_f32=None, std_f32=None) -> torch.Tensor:
    """On-disk embedding [C,H,W] (or a batch [b,C,H,W]) bf16 -> tokens [(h w), C] (decode_sample's rearrange,
    data_utils.py:152-155), normalised as below when statistics are given, widened to f32 (train_rvfm.py:112-114)."""
    x = x_chw_bf16
    tok = x.flatten(-2).transpose(-1, -2).contiguous()  # "c h w -> (h w) c"
    if mean_f32 is None:
        return tok.float()
    return normalize_feature_bf16(tok, mean_f32, std_f32)


def normalize_feature_bf16(x_bf16: torch.Tensor, mean_f32: torch.Tensor, std_f32: torch.Tensor) -> torch.Tensor:
    """x bf16 [HW,C]; stats are cast to bf16 (data_utils.py:374-379); (x-mean)/std in bf16 with a
    rounding after each op (data_utils.py:342-355); then .float() (train_rvfm.py:112-114)."""
    m = mean_f32.to(torch.bfloat16)
    s = std_f32.to(torch.bfloat16)
    return ((x_bf16 - m) / s).float()
