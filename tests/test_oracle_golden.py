"""CPU: the oracle (oracle/theia_oracle.py) against the golden vectors generated from the imported reference
(oracle/gen_golden.py -> tests/golden/*.npz).  This is what pins the checker; it runs everywhere (no GPU, no
/root/reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import theia_oracle as O

torch.set_num_threads(min(8, os.cpu_count() or 1))


def rel(a, b):
    return abs(a - b) / (abs(b) + 1e-30)


@pytest.mark.parametrize("name", ["g1_tiny_dinov2_b8", "g2_tiny_cdiv_b2", "g3_tiny_cddsv_b2", "g4_small_cddsv_b1", "g13_tiny_dinov2_cls_b2",
                                  "g15_nocls_tiny_dinov2_b2", "g16_reg_tiny_dinov2_b2"])
def test_oracle_matches_reference_goldens(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    bb, teachers, B = str(g["meta_backbone"]), [str(t) for t in g["meta_teachers"]], int(g["meta_B"])
    params = O.synth_params(bb, teachers, 0)
    images, targets = O.synth_images(B, 0), O.synth_targets(B, teachers, 1)
    with torch.no_grad():
        f = O.forward_feature(params, images, bb).numpy().reshape(-1)
    assert np.abs(f[g["feat_idx"]] - g["feat_val"]).max() / np.abs(g["feat_val"]).max() < 1e-5
    assert rel(np.abs(f.astype(np.float64)).sum(), float(g["feat_abssum"])) < 1e-6
    losses, main, grads, pred = O.train_step_grads(params, images, targets, bb, teachers, "cos_l1")
    assert rel(float(losses["mse_loss"]), float(g["mse_loss"])) < 1e-5
    assert rel(float(losses["cos_loss"]), float(g["cos_loss"])) < 1e-5
    assert rel(float(losses["l1_loss"]), float(g["l1_loss"])) < 1e-5
    assert rel(float(main), float(g["main_cos_l1"])) < 1e-5
    for ti, t in enumerate(teachers):
        p = pred[t].numpy().reshape(-1)
        assert np.abs(p[g[f"pred{ti}_idx"]] - g[f"pred{ti}_val"]).max() / np.abs(g[f"pred{ti}_val"]).max() < 1e-5
        assert rel(losses["cos_losses_per_model"][t], g["cos_pm"][ti]) < 1e-5
    for i, k in enumerate(str(n) for n in g["grad_names"]):
        gn = float(grads[k].double().pow(2).sum().sqrt())
        if "k_proj.bias" in k:  # exact gradient is identically zero (softmax shift invariance): rounding noise only
            assert gn < 1e-6
            continue
        assert rel(gn, g["gradnorm_cos_l1"][i]) < 1e-3, k  # fp32 summation-order noise on tiny norms


G17_CASES = (("deit", "facebook/deit-tiny-patch16-224"), ("nocls", "nocls-facebook/deit-tiny-patch16-224"), ("reg", "reg-facebook/deit-tiny-patch16-224"))


@pytest.mark.parametrize("tag,bb", G17_CASES)
def test_oracle_interpolated_position_embeddings_g17(golden_dir, tag, bb):
    """160x192 inputs, do_resize=False + interpolate_pos_encoding=True, on the reference (golden G17): features and the
    gradients that flow through the bicubic interpolation into position_embeddings (both interpolation flavours)."""
    g = np.load(os.path.join(golden_dir, "g17_interpolate_pos.npz"))
    params = {k: v.clone().requires_grad_(True) for k, v in O.synth_params(bb, O.TEACHER_SETS["dinov2"], 0).items()}
    z = O.forward_feature(params, torch.from_numpy(g["img"]), bb, interpolate_pos_encoding=True)
    assert tuple(z.shape) == tuple(g[f"{tag}_z_shape"])
    zz = z.detach().numpy().reshape(-1)
    assert np.abs(zz[g[f"{tag}_z_idx"]] - g[f"{tag}_z_val"]).max() / np.abs(g[f"{tag}_z_val"]).max() < 1e-5
    assert rel(np.abs(zz.astype(np.float64)).sum(), float(g[f"{tag}_z_abssum"])) < 1e-6
    w = torch.from_numpy(O._hash_uniform(z.numel(), 77).reshape(tuple(z.shape)).copy())
    (z * w).sum().backward()
    for short, k in (("pos", "backbone.model.embeddings.position_embeddings"), ("fc1", "backbone.model.layers.3.mlp.fc1.weight"),
                     ("patch", "backbone.model.embeddings.patch_embeddings.projection.weight")):
        gr = params[k].grad.numpy().reshape(-1)
        assert rel(float(np.sqrt((gr.astype(np.float64) ** 2).sum())), float(g[f"{tag}_g{short}_norm"])) < 1e-3, (tag, short)


def test_oracle_micro_ops_g10(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_micro_ops.npz"))

    def nhwc(a):
        return torch.from_numpy(a).permute(0, 2, 3, 1).contiguous()

    def nchw(t):
        return t.permute(0, 3, 1, 2).numpy()

    w, b = torch.from_numpy(g["w"]), torch.from_numpy(g["b"])
    assert np.abs(nchw(O.convT3x3(nhwc(g["x14"]), w, b, 1, 0, 0)) - g["convT_s1"]).max() < 5e-6
    assert np.abs(nchw(O.conv3x3_p1(nhwc(g["x16"]), w, b)) - g["conv_p1"]).max() < 5e-6
    assert np.abs(nchw(O.convT3x3(nhwc(g["x16"]), w, b, 2, 1, 0)) - g["convT_s2_p1"]).max() < 5e-6
    assert np.abs(nchw(O.convT3x3(nhwc(g["x31"]), w, b, 2, 0, 1)) - g["convT_s2_op1"]).max() < 5e-6
    ln = O.layernorm_chw(nhwc(g["x16"]), torch.from_numpy(g["ln_g"]), torch.from_numpy(g["ln_s"]))
    assert np.abs(nchw(ln) - g["ln_chw"]).max() < 5e-6
    xr = torch.from_numpy(g["xr"])
    assert np.abs(O._layernorm_rows(xr, torch.from_numpy(g["gr"]), torch.from_numpy(g["sr"]), 1e-12).numpy() - g["ln_row_eps1e12"]).max() < 5e-6
    assert np.abs(O._gelu_erf(xr).numpy() - g["gelu_erf"]).max() < 1e-6
    p, q = torch.from_numpy(g["lp"]), torch.from_numpy(g["lq"])
    L = O.get_loss({"t": p}, {"t": q})
    assert rel(float(L["mse_loss"]), float(g["mse"])) < 1e-6
    assert rel(float(L["l1_loss"]), float(g["smooth_l1"])) < 1e-6
    assert abs(float(L["cos_loss"]) - float(g["cos"])) < 1e-6


def test_oracle_tokens_layouts_and_lut_g6_g7(golden_dir):
    from PIL import Image
    g = np.load(os.path.join(golden_dir, "g6_g7_tokens_layouts.npz"))
    assert np.array_equal(O.preprocess_lut(), g["preproc_lut"])  # bit exact vs the HF processor run by the reference
    assert np.array_equal(O.preprocess_lut(True, False)[:, :224], g["preproc_lut_nonorm_0_223"])
    assert float(g["g7_bchw_maxdiff"]) == 0.0 and float(g["g7_pil_maxdiff"]) == 0.0 and float(g["g7_noresize_maxdiff"]) == 0.0
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    params = O.synth_params(bb, teachers, 0)
    images = O.synth_images(2, seed=3)
    with torch.no_grad():
        z = O.vit_forward(params, O.preprocess(images), bb)
        a = O.forward_feature(params, images, bb)
        assert torch.equal(a, O.forward_feature(params, images.permute(0, 3, 1, 2).contiguous(), bb))
        assert torch.equal(a, O.forward_feature(params, [Image.fromarray(images[i].numpy()) for i in range(2)], bb))
    zz = z.numpy().reshape(-1)
    assert np.abs(zz[g["z_idx"]] - g["z_val"]).max() / np.abs(g["z_val"]).max() < 1e-5
    for mode in ("mean_pooling", "max_pooling", "cls", "identity", None):
        y = O.handle_feature_output(z, mode, 0)
        key = "none" if mode is None else mode
        assert tuple(y.shape) == tuple(g[f"hfo_{key}_shape"])
        yy = y.numpy().reshape(-1)
        assert np.abs(yy[g[f"hfo_{key}_idx"]] - g[f"hfo_{key}_val"]).max() / np.abs(g[f"hfo_{key}_val"]).max() < 1e-5
    y = O.handle_feature_output(z, None, 3)
    assert tuple(y.shape) == tuple(g["hfo_none_disc3_shape"])
    with pytest.raises(NotImplementedError):
        O.handle_feature_output(z, "bogus", 0)


def test_oracle_feature_norm_bf16_g8(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_feature_norm_bf16.npz"))
    x = torch.from_numpy(g["x_bits"]).view(torch.bfloat16)
    y = O.normalize_feature_bf16(x, torch.from_numpy(g["mean"]), torch.from_numpy(g["std"])).numpy()
    assert np.array_equal(y, g["y"])


def test_oracle_feature_ingest_g11(golden_dir):
    """safetensors blob -> [C,H,W] bf16 -> (h w) c -> bf16 normalisation -> f32, against the reference's library calls."""
    from theia_amd.dataset import decode_feature
    g = np.load(os.path.join(golden_dir, "g11_feature_ingest.npz"))
    x = torch.from_numpy(g["x_bits"]).view(torch.bfloat16)
    dec = decode_feature(g["blob"].tobytes())["embedding"]
    assert dec.dtype == torch.bfloat16 and torch.equal(dec, x)
    y = O.ingest_feature_chw_bf16(dec, torch.from_numpy(g["mean"]), torch.from_numpy(g["std"])).numpy()
    assert np.array_equal(y, g["y"])
    assert np.array_equal(O.ingest_feature_chw_bf16(dec).numpy(), g["y_plain"])
    yb = O.ingest_feature_chw_bf16(torch.stack([dec, dec]), torch.from_numpy(g["mean"]), torch.from_numpy(g["std"])).numpy()
    assert yb.shape == (2,) + g["y"].shape and np.array_equal(yb[1], g["y"])


def test_dp_equivalence_g9(golden_dir):
    """Reference DDP(gloo, 2 ranks x b=2) == single process b=4 (what data parallelism must preserve)."""
    g = np.load(os.path.join(golden_dir, "g9_dp2_vs_single.npz"))
    a, b = g["gradnorm_ddp2"], g["gradnorm_single_b4"]
    for i, k in enumerate(str(n) for n in g["grad_names"]):
        if "k_proj.bias" in k:
            continue
        assert rel(a[i], b[i]) < 1e-4, k
