"""Generate golden fixtures from the REFERENCE itself (build container only; never shipped/run on the GPU box).

Imports the reference's own ``RobotVisionFM`` from /root/reference/src with the three shims of
SURVEY.md Appendix C (stub ``omegaconf``; local ``ViTModel(ViTConfig)`` instead of
``AutoModel.from_pretrained``; local ``ViTImageProcessorPil`` instead of
``AutoProcessor.from_pretrained``), loads this repo's deterministic synthetic weights into it,
runs forward_feature / forward / get_loss / backward on synthetic inputs and writes SMALL data
fixtures (scalars, sampled slices, checksums, grad norms) to ``tests/golden/*.npz``.

The fixtures hold data only -- no reference source text.  Usage (in the build container):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Cases: G1 tiny+dinov2 B=8; G2 tiny+cdiv B=2; G3 tiny+cddsv B=2; G4 small+cddsv B=1;
G5 base+cddsv B=1; G6 handle_feature_output modes; G7 input-layout equivalence; G8 bf16 feature
norm; G9 DP 2 ranks x b=2 vs 1 x b=4 (gloo DDP on the reference); G10 per-op micro goldens; G15 / G16 nocls- and reg-
students; G17 position-embedding interpolation on non-224 inputs (all three students).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import theia_oracle as O  # noqa: E402

REF_SRC = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    """Appendix C recipe."""
    if "omegaconf" not in sys.modules:
        om = types.ModuleType("omegaconf")

        class OmegaConf:  # noqa: D401 - stub
            @staticmethod
            def to_container(x, **kw):
                return dict(x)

        om.OmegaConf = OmegaConf
        om.DictConfig = dict
        sys.modules["omegaconf"] = om
    sys.path.insert(0, REF_SRC)
    import transformers
    from transformers import ViTConfig, ViTModel
    from transformers.models.vit.image_processing_pil_vit import ViTImageProcessorPil

    def fake_model_from_pretrained(name, *a, **k):
        D, nh, F = O.ARCH[name]
        return ViTModel(ViTConfig(hidden_size=D, num_attention_heads=nh, intermediate_size=F))

    def fake_model_from_config(cfg, *a, **k):
        return ViTModel(cfg)

    def fake_processor_from_pretrained(name, *a, **k):
        return ViTImageProcessorPil(image_mean=list(O.IMAGENET_MEAN), image_std=list(O.IMAGENET_STD),
                                    size={"height": 224, "width": 224}, do_resize=True)

    def fake_config_from_pretrained(name, *a, **k):
        D, nh, F = O.ARCH[name]
        return ViTConfig(hidden_size=D, num_attention_heads=nh, intermediate_size=F)

    transformers.AutoModel.from_pretrained = staticmethod(fake_model_from_pretrained)
    transformers.AutoModel.from_config = staticmethod(fake_model_from_config)
    transformers.AutoProcessor.from_pretrained = staticmethod(fake_processor_from_pretrained)
    transformers.AutoConfig.from_pretrained = staticmethod(fake_config_from_pretrained)
    # The reference's nocls- / reg- embeddings (backbones.py:26-209) were written against transformers 4.4x: they read
    # `self.config` of ViTEmbeddings and pass `interpolate_pos_encoding=` to ViTPatchEmbeddings.forward.  transformers 5.x has
    # neither (the flag only switched off the input-size check there).  Two compatibility shims, nothing else, let the
    # reference's own classes run unchanged:
    from transformers.models.vit import modeling_vit as MV
    if not getattr(MV.ViTEmbeddings, "_theia_compat", False):
        _init, _pf = MV.ViTEmbeddings.__init__, MV.ViTPatchEmbeddings.forward

        def emb_init(self, config, use_mask_token=False):
            _init(self, config, use_mask_token)
            self.config = config

        def patch_forward(self, pixel_values, interpolate_pos_encoding=False):
            return _pf(self, pixel_values)

        MV.ViTEmbeddings.__init__ = emb_init
        MV.ViTPatchEmbeddings.forward = patch_forward
        MV.ViTEmbeddings._theia_compat = True
    from theia.models.rvfm import RobotVisionFM
    from theia.foundation_models.common import get_model_feature_size
    from theia.models.utils import handle_feature_output
    return RobotVisionFM, get_model_feature_size, handle_feature_output


def build_reference(RobotVisionFM, get_model_feature_size, backbone, teachers, seed=0):
    torch.manual_seed(0)
    m = RobotVisionFM(
        backbone=backbone, pretrained=False, translator="lconv",
        translator_kwargs={"hidden_size_factor": 1.0},
        # "<teacher>_cls" heads as scripts/train/train_rvfm.py:238-246 sizes them
        target_feature_sizes={t: (get_model_feature_size(t[:-4], keep_spatial=True)[:1] if t.endswith("_cls")
                                  else get_model_feature_size(t, keep_spatial=True)) for t in teachers},
    )
    params = O.synth_params(backbone, teachers, seed)
    sd = m.state_dict()
    missing = [k for k in params if k not in sd]
    assert not missing, missing[:5]
    extra = [k for k in sd if k not in params and "pooler" not in k]
    assert not extra, extra[:5]
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    m.load_state_dict({k: v.clone() for k, v in params.items()}, strict=False)
    return m, params


def sample_idx(n: int, k: int, seed: int) -> np.ndarray:
    u = O._hash_uniform(k, 31337 + seed)
    return np.minimum(((u.astype(np.float64) + 1.0) * 0.5 * n).astype(np.int64), n - 1)


def run_case(name, backbone, teachers, B, RobotVisionFM, gmfs, with_grads=True, loss_kinds=("cos_l1", "mse")):
    print(f"[gen_golden] {name}: {backbone} x {len(teachers)} teachers, B={B}", flush=True)
    torch.set_num_threads(8)
    model, params = build_reference(RobotVisionFM, gmfs, backbone, teachers)
    images = O.synth_images(B, seed=0)
    targets = O.synth_targets(B, teachers, seed=1)
    fx = {"meta_backbone": np.array(backbone), "meta_teachers": np.array(list(teachers)), "meta_B": np.array(B)}

    model.eval()
    with torch.no_grad():
        feat = model.forward_feature(images)  # [B,196,D]
    f = feat.numpy().reshape(-1)
    idx = sample_idx(f.size, 256, 1)
    fx["feat_shape"] = np.array(feat.shape)
    fx["feat_idx"] = idx
    fx["feat_val"] = f[idx]
    fx["feat_sum"] = np.array(f.astype(np.float64).sum())
    fx["feat_abssum"] = np.array(np.abs(f.astype(np.float64)).sum())

    model.train()
    for kind in loss_kinds:
        model.zero_grad(set_to_none=True)
        pred = model(images)
        losses = model.get_loss(pred, targets)
        if kind == "mse":
            main = losses["mse_loss"]
        else:
            main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
        if kind == loss_kinds[0]:
            for ti, t in enumerate(teachers):
                p = pred[t].detach().numpy().reshape(-1)
                pi = sample_idx(p.size, 128, 100 + ti)
                fx[f"pred{ti}_shape"] = np.array(pred[t].shape)
                fx[f"pred{ti}_idx"] = pi
                fx[f"pred{ti}_val"] = p[pi]
                fx[f"pred{ti}_abssum"] = np.array(np.abs(p.astype(np.float64)).sum())
            fx["mse_loss"] = np.array(float(losses["mse_loss"]))
            fx["cos_loss"] = np.array(float(losses["cos_loss"]))
            fx["l1_loss"] = np.array(float(losses["l1_loss"]))
            fx["mse_pm"] = np.array([losses["mse_losses_per_model"][t] for t in teachers])
            fx["cos_pm"] = np.array([losses["cos_losses_per_model"][t] for t in teachers])
            fx["l1_pm"] = np.array([losses["l1_losses_per_model"][t] for t in teachers])
        fx[f"main_{kind}"] = np.array(float(main))
        if with_grads:
            main.backward()
            names = list(params.keys())
            sd_params = dict(model.named_parameters())
            gn = np.zeros(len(names))
            gsamp = np.zeros((len(names), 4), dtype=np.float32)
            gidx = np.zeros((len(names), 4), dtype=np.int64)
            for i, k in enumerate(names):
                g = sd_params[k].grad
                g = torch.zeros_like(sd_params[k]) if g is None else g
                gf = g.detach().numpy().reshape(-1)
                gn[i] = np.sqrt((gf.astype(np.float64) ** 2).sum())
                ii = sample_idx(gf.size, 4, 500 + i)
                gidx[i] = ii
                gsamp[i] = gf[ii]
            fx[f"grad_names"] = np.array(names)
            fx[f"gradnorm_{kind}"] = gn
            fx[f"gradidx_{kind}"] = gidx
            fx[f"gradval_{kind}"] = gsamp
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
    return model, params


def gen_g6_g7(RobotVisionFM, gmfs, hfo):
    print("[gen_golden] G6/G7", flush=True)
    backbone = "facebook/deit-tiny-patch16-224"
    teachers = O.TEACHER_SETS["dinov2"]
    model, params = build_reference(RobotVisionFM, gmfs, backbone, teachers)
    model.eval()
    images = O.synth_images(2, seed=3)
    fx = {}
    with torch.no_grad():
        z = model.backbone(images)  # [2,197,D]
        fx["z_abssum"] = np.array(np.abs(z.numpy().astype(np.float64)).sum())
        zi = sample_idx(z.numel(), 64, 9)
        fx["z_idx"] = zi
        fx["z_val"] = z.numpy().reshape(-1)[zi]
        for mode in ("mean_pooling", "max_pooling", "cls", "identity", None):
            y = hfo(z, mode, 0)
            key = "none" if mode is None else mode
            fx[f"hfo_{key}_shape"] = np.array(y.shape)
            yy = y.numpy().reshape(-1)
            yi = sample_idx(yy.size, 32, 10)
            fx[f"hfo_{key}_idx"] = yi
            fx[f"hfo_{key}_val"] = yy[yi]
        # discard tokens variant
        y = hfo(z, None, 3)
        fx["hfo_none_disc3_shape"] = np.array(y.shape)
        fx["hfo_none_disc3_abssum"] = np.array(np.abs(y.numpy().astype(np.float64)).sum())
        # G7 layout equivalence on the reference: BHWC == BCHW == list of PIL; do_resize True/False
        from PIL import Image
        a = model.forward_feature(images)
        bchw = model.forward_feature(images.permute(0, 3, 1, 2).contiguous())
        pil = model.forward_feature([Image.fromarray(images[i].numpy()) for i in range(2)])
        nr = model.forward_feature(images, do_resize=False)
        fx["g7_bchw_maxdiff"] = np.array(float((a - bchw).abs().max()))
        fx["g7_pil_maxdiff"] = np.array(float((a - pil).abs().max()))
        fx["g7_noresize_maxdiff"] = np.array(float((a - nr).abs().max()))
        # preprocessing LUT as the reference's processor computes it
        ramp = torch.arange(256, dtype=torch.uint8).view(1, 1, 256, 1).expand(1, 224, 256, 3)[:, :, :224].contiguous()
        ramp2 = (torch.arange(224, dtype=torch.int32) + 32).to(torch.uint8).view(1, 1, 224, 1).expand(1, 224, 224, 3).contiguous()
        pv = model.backbone.processor(ramp, return_tensors="pt", do_resize=True)["pixel_values"]
        pv2 = model.backbone.processor(ramp2, return_tensors="pt", do_resize=True)["pixel_values"]
        lut = np.zeros((3, 256), dtype=np.float32)
        lut[:, :224] = pv[0, :, 0, :].numpy()
        lut[:, 32:256] = pv2[0, :, 0, :].numpy()
        fx["preproc_lut"] = lut
        pv_nonorm = model.backbone.processor(ramp, return_tensors="pt", do_resize=True, do_normalize=False)["pixel_values"]
        fx["preproc_lut_nonorm_0_223"] = pv_nonorm[0, :, 0, :].numpy()
    np.savez_compressed(os.path.join(OUT, "g6_g7_tokens_layouts.npz"), **fx)


def gen_g12(RobotVisionFM, gmfs):
    """Non-224 inputs through the REFERENCE model's own processor and backbone (backbones.py:337-339, do_resize=True):
    the resized uint8 image the processor produces (do_rescale=False, do_normalize=False returns it), the full pixel_values,
    and forward_feature of the reference on those inputs."""
    print("[gen_golden] G12", flush=True)
    backbone = "facebook/deit-tiny-patch16-224"
    teachers = O.TEACHER_SETS["dinov2"]
    model, params = build_reference(RobotVisionFM, gmfs, backbone, teachers)
    model.eval()
    proc = model.backbone.processor
    fx = {}
    rows = np.arange(0, 224, 7)
    cases = {"up_hwc": (2, 50, 37, True), "down_chw": (1, 150, 131, False), "xonly_hwc": (1, 224, 120, True), "yonly_chw": (1, 90, 224, False)}
    for name, (b, hh, ww, cl) in cases.items():
        img = torch.from_numpy((O._hash_uniform(b * hh * ww * 3, 300 + hh) * 0.5 + 0.5).reshape(b, hh, ww, 3) * 255.999).to(torch.uint8)
        x = img if cl else img.permute(0, 3, 1, 2).contiguous()
        fx[f"{name}_img"] = x.numpy()
        with torch.no_grad():
            raw = proc(x, return_tensors="pt", do_resize=True, do_rescale=False, do_normalize=False)["pixel_values"]  # [b,3,224,224]
            pv = proc(x, return_tensors="pt", do_resize=True)["pixel_values"]
            z = model.forward_feature(x)
        r8 = raw.permute(0, 2, 3, 1).round().to(torch.uint8).numpy()  # HWC
        assert np.array_equal(r8.astype(np.float32), raw.permute(0, 2, 3, 1).numpy())
        fx[f"{name}_resized_rows"] = r8[:, rows]
        fx[f"{name}_resized_sum"] = np.array(r8.astype(np.int64).sum())
        fx[f"{name}_pv_rows"] = pv.permute(0, 2, 3, 1).numpy()[:, rows[::4]]
        zi = sample_idx(z.numel(), 64, 13)
        fx[f"{name}_z_idx"] = zi
        fx[f"{name}_z_val"] = z.numpy().reshape(-1)[zi]
        fx[f"{name}_z_abssum"] = np.array(np.abs(z.numpy().astype(np.float64)).sum())
    fx["rows"] = rows
    np.savez_compressed(os.path.join(OUT, "g12_resize_processor.npz"), **fx)


def gen_g14(RobotVisionFM, gmfs):
    """The DeiT processor configuration of the hub checkpoints (preprocessor_config.json of facebook/deit-*-patch16-224:
    resize 256 bicubic, center-crop 224, ImageNet mean/std; transformers' DeiTImageProcessor) plugged into the reference
    model in place of the offline shim's ViT processor: resized+cropped uint8 rows, pixel_values rows, forward_feature."""
    print("[gen_golden] G14", flush=True)
    from transformers.models.deit.image_processing_pil_deit import DeiTImageProcessorPil
    backbone = "facebook/deit-tiny-patch16-224"
    teachers = O.TEACHER_SETS["dinov2"]
    model, params = build_reference(RobotVisionFM, gmfs, backbone, teachers)
    model.eval()
    proc = DeiTImageProcessorPil(image_mean=list(O.IMAGENET_MEAN), image_std=list(O.IMAGENET_STD))
    assert proc.size["height"] == 256 and proc.crop_size["height"] == 224 and int(proc.resample) == 3 and proc.do_center_crop
    model.backbone.processor = proc
    fx = {}
    rows = np.arange(0, 224, 7)
    for name, (b, hh, ww) in {"in224": (2, 224, 224), "in300x260": (1, 300, 260)}.items():
        img = torch.from_numpy((O._hash_uniform(b * hh * ww * 3, 500 + hh) * 0.5 + 0.5).reshape(b, hh, ww, 3) * 255.999).to(torch.uint8)
        if name == "in224":
            img = O.synth_images(b, 0)
        else:
            fx[f"{name}_img"] = img.numpy()
        with torch.no_grad():
            raw = proc(img, return_tensors="pt", do_rescale=False, do_normalize=False)["pixel_values"]
            pv = proc(img, return_tensors="pt")["pixel_values"]
            z = model.forward_feature(img)
        r8 = raw.permute(0, 2, 3, 1).to(torch.uint8).numpy()
        fx[f"{name}_u8_rows"] = r8[:, rows]
        fx[f"{name}_u8_sum"] = np.array(r8.astype(np.int64).sum())
        fx[f"{name}_pv_rows"] = pv.permute(0, 2, 3, 1).numpy()[:, rows[::4]]
        zi = sample_idx(z.numel(), 64, 17)
        fx[f"{name}_z_idx"] = zi
        fx[f"{name}_z_val"] = z.numpy().reshape(-1)[zi]
        fx[f"{name}_z_abssum"] = np.array(np.abs(z.numpy().astype(np.float64)).sum())
    fx["rows"] = rows
    np.savez_compressed(os.path.join(OUT, "g14_deit_processor.npz"), **fx)


def gen_g17(RobotVisionFM, gmfs):
    """interpolate_pos_encoding on non-224 input with do_resize=False (backbones.py:314-341 -> HF modeling_vit.py:89-127 for
    DeiT; the reference's own interpolate_pos_encoding for nocls- / reg-, backbones.py:39-69,146-177): forward_feature of the
    reference on 160x192 images, and the gradient that flows back through the interpolation into position_embeddings."""
    print("[gen_golden] G17", flush=True)
    teachers = O.TEACHER_SETS["dinov2"]
    fx = {}
    hh, ww, b = 160, 192, 2
    img = torch.from_numpy((O._hash_uniform(b * hh * ww * 3, 700) * 0.5 + 0.5).reshape(b, hh, ww, 3) * 255.999).to(torch.uint8)
    fx["img"] = img.numpy()
    for tag, bb in (("deit", "facebook/deit-tiny-patch16-224"), ("nocls", "nocls-facebook/deit-tiny-patch16-224"),
                    ("reg", "reg-facebook/deit-tiny-patch16-224")):
        model, params = build_reference(RobotVisionFM, gmfs, bb, teachers)
        model.train()
        z = model.forward_feature(img, do_resize=False, interpolate_pos_encoding=True)
        zz = z.detach().numpy().reshape(-1)
        zi = sample_idx(zz.size, 64, 23)
        fx[f"{tag}_z_shape"] = np.array(z.shape)
        fx[f"{tag}_z_idx"] = zi
        fx[f"{tag}_z_val"] = zz[zi]
        fx[f"{tag}_z_abssum"] = np.array(np.abs(zz.astype(np.float64)).sum())
        w = torch.from_numpy(O._hash_uniform(z.numel(), 77).reshape(tuple(z.shape)).copy())
        (z * w).sum().backward()
        sdp = dict(model.named_parameters())
        for short, k in (("pos", "backbone.model.embeddings.position_embeddings"), ("fc1", "backbone.model.layers.3.mlp.fc1.weight"),
                         ("patch", "backbone.model.embeddings.patch_embeddings.projection.weight")):
            g = sdp[k].grad.detach().numpy().reshape(-1)
            gi = sample_idx(g.size, 16, 29)
            fx[f"{tag}_g{short}_norm"] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            fx[f"{tag}_g{short}_idx"] = gi
            fx[f"{tag}_g{short}_val"] = g[gi]
    np.savez_compressed(os.path.join(OUT, "g17_interpolate_pos.npz"), **fx)


def gen_g8():
    print("[gen_golden] G8", flush=True)
    sys.path.insert(0, REF_SRC)
    # reference normalize_feature is `(x - mean) / std` on bf16 tensors (data_utils.py:342-355); importing
    # data_utils needs webdataset/cv2 (absent), so the one-line expression is evaluated by torch directly
    # on bf16 tensors prepared exactly as data_utils.py:374-379 prepares them.
    HW, C = 64, 48
    x = torch.from_numpy(O._hash_uniform(HW * C, 77).reshape(HW, C) * 6.0).to(torch.bfloat16)
    mean = torch.from_numpy(O._hash_uniform(C, 78) * 0.5)
    std = torch.from_numpy(np.abs(O._hash_uniform(C, 79)) * 2.0 + 0.25)
    y = ((x - mean.to(torch.bfloat16)) / std.to(torch.bfloat16)).float()
    np.savez_compressed(os.path.join(OUT, "g8_feature_norm_bf16.npz"),
                        x_bits=x.view(torch.int16).numpy(), mean=mean.numpy(), std=std.numpy(), y=y.numpy())


def gen_g11():
    """Teacher-feature ingest: the reference's decode_sample branch for ".safetensors" keys (data_utils.py:150-155:
    sft_load -> rearrange "c h w -> (h w) c" -> feature_transform) followed by .float() (train_rvfm.py:112-114).
    data_utils itself cannot be imported here (webdataset / cv2 are absent), so its three lines are evaluated with the same
    library calls (safetensors.torch.load, einops.rearrange, the normalize_feature expression on bf16 tensors)."""
    print("[gen_golden] G11", flush=True)
    from einops import rearrange
    from safetensors.torch import load as sft_load, save as sft_save
    C, H, W = 40, 6, 5
    x = torch.from_numpy(O._hash_uniform(C * H * W, 91).reshape(C, H, W) * 5.0).to(torch.bfloat16)
    mean = torch.from_numpy(O._hash_uniform(C, 92) * 0.5)
    std = torch.from_numpy(np.abs(O._hash_uniform(C, 93)) * 2.0 + 0.25)
    blob = sft_save({"embedding": x})
    emb = rearrange(sft_load(blob)["embedding"], "c h w -> (h w) c")
    y = ((emb - mean.to(torch.bfloat16)) / std.to(torch.bfloat16)).float()
    np.savez_compressed(os.path.join(OUT, "g11_feature_ingest.npz"), blob=np.frombuffer(blob, dtype=np.uint8),
                        x_bits=x.view(torch.int16).numpy(), mean=mean.numpy(), std=std.numpy(), y=y.numpy(),
                        y_plain=emb.float().numpy())


def gen_g10():
    """Per-op micro goldens straight from torch.nn (what the reference's modules call)."""
    print("[gen_golden] G10", flush=True)
    import torch.nn as nn
    import torch.nn.functional as F
    fx = {}
    C = 8

    def h(shape, seed, scale=1.0):
        return torch.from_numpy((O._hash_uniform(int(np.prod(shape)), seed) * scale).reshape(shape))

    x14 = h((2, C, 14, 14), 1)
    w = h((C, C, 3, 3), 2, 0.3)
    b = h((C,), 3, 0.1)
    fx["x14"] = x14.numpy(); fx["w"] = w.numpy(); fx["b"] = b.numpy()
    fx["convT_s1"] = F.conv_transpose2d(x14, w, b, stride=1).numpy()  # 14->16
    x16 = h((2, C, 16, 16), 4)
    fx["x16"] = x16.numpy()
    fx["conv_p1"] = F.conv2d(x16, w, b, padding=1).numpy()
    fx["convT_s2_p1"] = F.conv_transpose2d(x16, w, b, stride=2, padding=1).numpy()  # 31
    x31 = h((2, C, 31, 31), 5)
    fx["x31"] = x31.numpy()
    fx["convT_s2_op1"] = F.conv_transpose2d(x31, w, b, stride=2, output_padding=1).numpy()  # 64
    g = h((C, 16, 16), 6, 0.2) + 1.0
    s = h((C, 16, 16), 7, 0.2)
    fx["ln_g"] = g.numpy(); fx["ln_s"] = s.numpy()
    fx["ln_chw"] = F.layer_norm(x16, (C, 16, 16), g, s, 1e-5).numpy()
    xr = h((5, 24), 8, 2.0)
    gr = h((24,), 9, 0.2) + 1.0
    sr = h((24,), 10, 0.2)
    fx["xr"] = xr.numpy(); fx["gr"] = gr.numpy(); fx["sr"] = sr.numpy()
    fx["ln_row_eps1e12"] = F.layer_norm(xr, (24,), gr, sr, 1e-12).numpy()
    fx["gelu_erf"] = F.gelu(xr).numpy()
    p = h((3, 10, 6), 11, 2.0)
    q = h((3, 10, 6), 12, 2.0)
    fx["lp"] = p.numpy(); fx["lq"] = q.numpy()
    fx["smooth_l1"] = np.array(float(nn.SmoothL1Loss()(p, q)))
    fx["mse"] = np.array(float(nn.MSELoss()(p, q)))
    pn = F.normalize(p.flatten(start_dim=1), dim=1, p=2)
    qn = F.normalize(q.flatten(start_dim=1), dim=1, p=2)
    fx["cos"] = np.array(float(nn.CosineEmbeddingLoss()(pn, qn, torch.ones(3, dtype=torch.int))))
    np.savez_compressed(os.path.join(OUT, "g10_micro_ops.npz"), **fx)


def _ddp_worker(rank, world, backbone, teachers, port, ret):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    RobotVisionFM, gmfs, _ = import_reference()
    model, params = build_reference(RobotVisionFM, gmfs, backbone, teachers)
    ddp = DDP(model)
    B = 4
    images = O.synth_images(B, seed=0)[rank * 2:(rank + 1) * 2]
    targets = {t: v[rank * 2:(rank + 1) * 2] for t, v in O.synth_targets(B, teachers, seed=1).items()}
    pred = ddp(images)
    losses = model.get_loss(pred, targets)
    main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
    main.backward()
    if rank == 0:
        names = list(params.keys())
        sdp = dict(model.named_parameters())
        ret["gn"] = np.array([float(sdp[k].grad.double().pow(2).sum().sqrt()) for k in names])
        ret["names"] = names
        ret["main_rank0"] = float(main)
    dist.barrier()
    dist.destroy_process_group()


def gen_g9(RobotVisionFM, gmfs):
    print("[gen_golden] G9 (DDP gloo 2 ranks)", flush=True)
    import torch.multiprocessing as mp
    backbone = "facebook/deit-tiny-patch16-224"
    teachers = O.TEACHER_SETS["cdiv"]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, backbone, teachers, 29533, ret), nprocs=2, join=True)
    # single-process B=4
    model, params = build_reference(RobotVisionFM, gmfs, backbone, teachers)
    images = O.synth_images(4, seed=0)
    targets = O.synth_targets(4, teachers, seed=1)
    pred = model(images)
    losses = model.get_loss(pred, targets)
    main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
    main.backward()
    sdp = dict(model.named_parameters())
    names = list(params.keys())
    gn1 = np.array([float(sdp[k].grad.double().pow(2).sum().sqrt()) for k in names])
    np.savez_compressed(os.path.join(OUT, "g9_dp2_vs_single.npz"), grad_names=np.array(names),
                        gradnorm_ddp2=np.array(ret["gn"]), gradnorm_single_b4=gn1,
                        main_single_b4=np.array(float(main)), main_rank0_b2=np.array(ret["main_rank0"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    which = set(sys.argv[1:])
    RobotVisionFM, gmfs, hfo = import_reference()

    def want(n):
        return not which or n in which

    T = O.TEACHER_SETS
    if want("g1"):
        run_case("g1_tiny_dinov2_b8", "facebook/deit-tiny-patch16-224", T["dinov2"], 8, RobotVisionFM, gmfs)
    if want("g2"):
        run_case("g2_tiny_cdiv_b2", "facebook/deit-tiny-patch16-224", T["cdiv"], 2, RobotVisionFM, gmfs)
    if want("g3"):
        run_case("g3_tiny_cddsv_b2", "facebook/deit-tiny-patch16-224", T["cddsv"], 2, RobotVisionFM, gmfs)
    if want("g4"):
        run_case("g4_small_cddsv_b1", "facebook/deit-small-patch16-224", T["cddsv"], 1, RobotVisionFM, gmfs,
                 loss_kinds=("cos_l1",))
    if want("g5"):
        run_case("g5_base_cddsv_b1", "facebook/deit-base-patch16-224", T["cddsv"], 1, RobotVisionFM, gmfs,
                 loss_kinds=("cos_l1",))
    if want("g13"):  # CLS-token distillation heads (train_rvfm.py distill_cls): a spatial head + two "_cls" heads
        run_case("g13_tiny_dinov2_cls_b2", "facebook/deit-tiny-patch16-224",
                 ["facebook/dinov2-large", "facebook/dinov2-large_cls", "openai/clip-vit-large-patch14_cls"], 2, RobotVisionFM, gmfs)
    if want("g15"):  # student without CLS token (backbones.py:344-416)
        run_case("g15_nocls_tiny_dinov2_b2", "nocls-facebook/deit-tiny-patch16-224", T["dinov2"], 2, RobotVisionFM, gmfs)
    if want("g16"):  # student with 7 register tokens (backbones.py:419-503)
        run_case("g16_reg_tiny_dinov2_b2", "reg-facebook/deit-tiny-patch16-224", T["dinov2"], 2, RobotVisionFM, gmfs)
    if want("g17"):
        gen_g17(RobotVisionFM, gmfs)
    if want("g6"):
        gen_g6_g7(RobotVisionFM, gmfs, hfo)
    if want("g8"):
        gen_g8()
    if want("g12"):
        gen_g12(RobotVisionFM, gmfs)
    if want("g14"):
        gen_g14(RobotVisionFM, gmfs)
    if want("g10"):
        gen_g10()
    if want("g11"):
        gen_g11()
    if want("g9"):
        gen_g9(RobotVisionFM, gmfs)
    print("[gen_golden] done")


if __name__ == "__main__":
    main()
