"""End-to-end GPU parity: RobotVisionFM (HIP engine) vs golden fixtures produced by the imported reference
(tests/golden/*.npz, generator: oracle/gen_golden.py) and vs the CPU oracle on fresh inputs.

fp32 mode must meet the north-star tolerance: features / losses within 1e-4 relative; per-parameter gradient norms
within 1e-3 relative (k_proj.bias gradients are identically zero in exact arithmetic -> absolute tolerance).
bf16 mode (throughput path) is checked at its own stated tolerance against the fp32 goldens.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import theia_oracle as O  # noqa: E402  (checker only)

CASES = {
    "g1": "g1_tiny_dinov2_b8.npz",
    "g2": "g2_tiny_cdiv_b2.npz",
    "g3": "g3_tiny_cddsv_b2.npz",
    "g4": "g4_small_cddsv_b1.npz",
    "g5": "g5_base_cddsv_b1.npz",
    "g13": "g13_tiny_dinov2_cls_b2.npz",  # a spatial head + two CLS-token heads (distill_cls)
}


def build(backbone, teachers, precision, seed=0):
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.foundation_models.common import get_model_feature_size
    m = RobotVisionFM(backbone=backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: (get_model_feature_size(t[:-4], keep_spatial=True)[:1] if t.endswith("_cls")
                                                else get_model_feature_size(t, keep_spatial=True)) for t in teachers},
                      precision=precision)
    params = O.synth_params(backbone, teachers, seed)
    missing, unexpected = m.load_state_dict(params, strict=True), None
    return m.to("cuda:0"), params


def load_case(golden_dir, key):
    g = np.load(os.path.join(golden_dir, CASES[key]))
    return g, str(g["meta_backbone"]), [str(t) for t in g["meta_teachers"]], int(g["meta_B"])


def rel(a, b):
    return abs(a - b) / (abs(b) + 1e-30)


@pytest.mark.parametrize("key", ["g1", "g2", "g3", "g4", "g5", "g13"])
def test_fp32_matches_reference_goldens(golden_dir, key):
    g, bb, teachers, B = load_case(golden_dir, key)
    model, _ = build(bb, teachers, "fp32")
    check_against_golden(model, g, teachers, B, key)


def check_against_golden(model, g, teachers, B, key):
    """forward_feature, forward, the three losses + per-teacher values, main loss, every gradient norm and sampled gradient values of
    an fp32 model against one reference golden"""
    images = O.synth_images(B, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}
    # forward_feature (reference models/rvfm.py:94-113)
    with torch.no_grad():
        feat = model.forward_feature(images)
    assert tuple(feat.shape) == tuple(g["feat_shape"])
    f = feat.float().cpu().numpy().reshape(-1)
    scale = np.abs(g["feat_val"]).max()
    assert np.abs(f[g["feat_idx"]] - g["feat_val"]).max() / scale < 1e-4
    assert rel(np.abs(f.astype(np.float64)).sum(), float(g["feat_abssum"])) < 1e-5
    # forward + losses + backward (train_rvfm.py:116-125)
    model.train()
    pred = model(images)
    assert list(pred.keys()) == teachers
    for ti, t in enumerate(teachers):
        p = pred[t].detach().float().cpu().numpy().reshape(-1)
        assert tuple(pred[t].shape) == tuple(g[f"pred{ti}_shape"])
        assert np.abs(p[g[f"pred{ti}_idx"]] - g[f"pred{ti}_val"]).max() / np.abs(g[f"pred{ti}_val"]).max() < 1e-4
    losses = model.get_loss(pred, targets)
    assert rel(float(losses["mse_loss"]), float(g["mse_loss"])) < 1e-4
    assert rel(float(losses["cos_loss"]), float(g["cos_loss"])) < 1e-4
    assert rel(float(losses["l1_loss"]), float(g["l1_loss"])) < 1e-4
    for ti, t in enumerate(teachers):
        assert rel(losses["mse_losses_per_model"][t], g["mse_pm"][ti]) < 1e-4
        assert rel(losses["cos_losses_per_model"][t], g["cos_pm"][ti]) < 1e-4
        assert rel(losses["l1_losses_per_model"][t], g["l1_pm"][ti]) < 1e-4
    main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
    assert rel(float(main), float(g["main_cos_l1"])) < 1e-4
    main.backward()
    sdp = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    gn_ref = g["gradnorm_cos_l1"]
    worst = 0.0
    for i, k in enumerate(names):
        gr = sdp[k].grad
        assert gr is not None, k
        gn = float(gr.double().pow(2).sum().sqrt())
        if "k_proj.bias" in k:
            assert gn < 1e-6 * max(1.0, gn_ref.max()), (k, gn)
            continue
        r = rel(gn, gn_ref[i])
        worst = max(worst, r)
        assert r < 1e-3, (k, gn, gn_ref[i])
        gv = gr.detach().float().cpu().numpy().reshape(-1)[g["gradidx_cos_l1"][i]]
        # 4 sampled gradient values per tensor, error in units of the tensor's RMS gradient.  Tight level 3e-3 (measured worst
        # case of pure f32 summation-order noise: 3e-3 on DeiT-base's patch-embedding weight, the end of the 12-layer backward
        # chain).  ONE of the four may sit at the loose level 3e-2: a ReLU pre-activation of the translator heads within ~1e-7
        # of zero takes the other branch under a different f32 summation order (~0.3 such units per 3M-element map), which
        # moves the few gradient entries fed by that unit by ~1/sqrt(9C) of their RMS (seen: 1.4e-2 on one LN-affine entry of
        # golden G5) -- a discontinuity of the function, not of the kernels.  The norm check above is the 1e-3 gate.
        errs = np.sort(np.abs(gv - g["gradval_cos_l1"][i]) / (gn_ref[i] / np.sqrt(gr.numel()) + 1e-30))
        assert errs[-1] <= 3e-2 and errs[-2] <= 3e-3, (k, errs)
    print(f"[{key}] worst grad-norm rel err {worst:.2e}")


@pytest.mark.parametrize("key", ["g1", "g3"])
def test_fp32_mse_main_loss(golden_dir, key):
    g, bb, teachers, B = load_case(golden_dir, key)
    model, _ = build(bb, teachers, "fp32")
    images = O.synth_images(B, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}
    pred = model(images)
    losses = model.get_loss(pred, targets)
    assert rel(float(losses["mse_loss"]), float(g["main_mse"])) < 1e-4
    losses["mse_loss"].backward()
    sdp = dict(model.named_parameters())
    for i, k in enumerate([str(n) for n in g["grad_names"]]):
        if "k_proj.bias" in k:
            continue
        gn = float(sdp[k].grad.double().pow(2).sum().sqrt())
        assert rel(gn, g["gradnorm_mse"][i]) < 1e-3, k


@pytest.mark.parametrize("key", ["g1", "g3"])
def test_bf16_mode_tracks_fp32(golden_dir, key):
    """Throughput path: bf16 operands / f32 accumulate.  Tolerances: losses 2e-2 rel, per-parameter gradient
    cosine similarity vs the fp32 oracle gradients > 0.98 for the large matrices."""
    g, bb, teachers, B = load_case(golden_dir, key)
    model, params = build(bb, teachers, "bf16")
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, teachers, 1)
    targets = {t: v.to("cuda:0") for t, v in tcpu.items()}
    pred = model(images)
    losses = model.get_loss(pred, targets)
    assert rel(float(losses["mse_loss"]), float(g["mse_loss"])) < 2e-2
    assert rel(float(losses["cos_loss"]), float(g["cos_loss"])) < 2e-2
    assert rel(float(losses["l1_loss"]), float(g["l1_loss"])) < 2e-2
    (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    _, _, grads, _ = O.train_step_grads(params, images, tcpu, bb, teachers, "cos_l1")
    sdp = dict(model.named_parameters())
    for k, gref in grads.items():
        if gref.numel() < 4096 or "k_proj" in k:
            continue
        a = sdp[k].grad.detach().float().cpu().reshape(-1).double()
        b = gref.reshape(-1).double()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.98, (k, cos)


class tile_hint:
    """every theia_gemm_nt launch inside the block requests this tile (ops.GEMM_TILE_HINT)"""

    def __init__(self, tile):
        self.tile = tile

    def __enter__(self):
        from theia_amd import ops
        self.prev, ops.GEMM_TILE_HINT = ops.GEMM_TILE_HINT, self.tile

    def __exit__(self, *a):
        from theia_amd import ops
        ops.GEMM_TILE_HINT = self.prev


def _grad_agreement(model, grads_ref, min_numel=4096):
    """(worst cosine, worst |norm ratio - 1|, name of the worst) over the parameters with >= min_numel elements; the smaller tensors
    (biases, LayerNorm rows: a cosine over a few hundred entries is itself noisy) enter through the norm-relative error
    |a - b| / |b|, returned as the 4th value with its tensor's name.  k_proj gradients are excluded: the key bias gradient is
    identically zero in exact arithmetic (softmax is invariant to a shift of the keys) and the key weight gradient inherits that
    cancellation."""
    worst_cos, worst_nr, who = 1.0, 0.0, ""
    worst_small, who_small = 0.0, ""
    for k, p in model.named_parameters():
        gref = grads_ref[k]
        if "k_proj" in k:
            continue
        a = p.grad.detach().float().cpu().reshape(-1).double()
        b = gref.detach().float().cpu().reshape(-1).double()
        if gref.numel() < min_numel:
            e = float((a - b).norm() / (b.norm() + 1e-30))
            if e > worst_small:
                worst_small, who_small = e, k
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        nr = abs(float(a.norm() / (b.norm() + 1e-30)) - 1.0)
        if cos < worst_cos:
            worst_cos, who = cos, k
        worst_nr = max(worst_nr, nr)
    _grad_agreement.small = (worst_small, who_small)
    return worst_cos, worst_nr, who


def test_bf16_base_cddsv_on_the_pingpong_kernels_vs_oracle():
    """BASELINE config 3 in its stated mode: DeiT-base + 5 teachers (cddsv), bf16.  At B = 8 the library would pick 128-wide
    tiles, so every GEMM is forced onto the 256x256 ping-pong kernel the per-GPU-batch-128 bench runs on (ragged tiles in M:
    8*197 = 6.16 tiles; all row maps of the translator heads).  Checked against the CPU oracle's fp32 losses and gradients.
    Tolerances (bf16 operands / activations, f32 accumulate): losses 2e-2 rel, per-tensor gradient cosine > 0.99 and
    gradient-norm ratio within 2.5 % for tensors of >= 4096 elements, norm-relative error < 0.2 for the smaller ones."""
    bb, teachers, B = "facebook/deit-base-patch16-224", O.TEACHER_SETS["cddsv"], 8
    model, params = build(bb, teachers, "bf16")
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, teachers, 1)
    targets = {t: v.to("cuda:0") for t, v in tcpu.items()}
    with tile_hint(256256):
        pred = model(images)
        losses = model.get_loss(pred, targets)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    ref_losses, _, grads, _ = O.train_step_grads(params, images, tcpu, bb, teachers, "cos_l1")
    for k in ("mse_loss", "cos_loss", "l1_loss"):
        assert rel(float(losses[k]), float(ref_losses[k])) < 2e-2, (k, float(losses[k]), float(ref_losses[k]))
    cos, nr, who = _grad_agreement(model, grads)
    small, who_small = _grad_agreement.small
    print(f"[base bf16 pp] worst gradient cosine {cos:.5f} ({who}), worst norm-ratio error {nr:.4f}; "
          f"tensors < 4096 elements: worst |a-b|/|b| {small:.4f} ({who_small})")
    assert cos > 0.99 and nr < 2.5e-2, (cos, nr, who)  # measured: 0.9933 / 0.0103 -- the bf16 noise floor of this network
    assert small < 0.2, (small, who_small)           # measured: 0.112


def test_bf16_bench_dispatch_agrees_with_the_2stage_kernels():
    """DeiT-base + cddsv at B = 64, where the library's own dispatch puts the ViT and head GEMMs on the persistent ping-pong kernel
    (as at the bench's B = 128): the same step with every GEMM forced onto the 2-stage 128x128 kernel (the kernel the small-shape
    oracle tests run by default) must give the same losses and gradients up to bf16 rounding noise.

    What "bf16 rounding noise" is, measured (round 3): the ping-pong kernel starts its f32 accumulation from the bias / residual row
    instead of adding them last, so each GEMM's output differs from the 2-stage kernel's by ONE bf16 ulp in ~1e-4 of its elements
    (tests/test_ops_gpu.py::test_pingpong_matches_2stage_up_to_rare_ulp_flips) -- and 12 layers of LayerNorm / attention / residual
    later the two executions have decorrelated to the bf16 noise floor: ~40 % of the elements of a head activation differ in their
    last bit, losses agree to < 2e-3, and the most noise-sensitive gradients (the LayerNorm[C,16,16] affine in front of the first
    head convolution: sums over only B samples) agree to cosine 0.9938.  (With bit-identical accumulation order, as in round 2, the
    same comparison gave 0.9998: that figure measured identical arithmetic, not bf16 accuracy.)  Each execution is deterministic."""
    from theia_amd import _native as N
    bb, teachers, B = "facebook/deit-base-patch16-224", O.TEACHER_SETS["cddsv"], 64
    assert N.lib().theia_gemm_nt_tile(B * 197, 768, N.BF16) == 256256 and N.lib().theia_gemm_nt_tile(B * 256, 768, N.BF16) == 256256
    model, _ = build(bb, teachers, "bf16")
    images = O.synth_images(B, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}

    def step():
        model.zero_grad(set_to_none=True)
        losses = model.get_loss(model(images), targets)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
        return losses

    la = step()
    ga = {k: p.grad.clone() for k, p in model.named_parameters()}
    la2 = step()
    assert all(torch.equal(ga[k], p.grad) for k, p in model.named_parameters()) and float(la["cos_loss"]) == float(la2["cos_loss"])  # deterministic
    with tile_hint(128128):
        lb = step()
    for k in ("mse_loss", "cos_loss", "l1_loss"):
        assert rel(float(la[k]), float(lb[k])) < 2e-3, (k, float(la[k]), float(lb[k]))
    cos, nr, who = _grad_agreement(model, ga)
    print(f"[base bf16 B=64 auto vs 128x128] worst gradient cosine {cos:.6f} ({who}), worst norm-ratio error {nr:.5f}")
    assert cos > 0.99 and nr < 1.5e-2, (cos, nr, who)


def test_cu_reservation_during_backward_changes_the_schedule_not_the_result(monkeypatch):
    """What TheiaDataParallel does at N > 1 (theia_amd/parallel.py), exercised on one GPU: from the first completed gradient bucket
    to the end of the backward pass the GEMM planners leave 16 CUs to RCCL (persistent NT grid 240, weight-gradient splits for 240
    CUs).  The NT kernels compute every output element in the same order whatever the grid, so losses and data gradients are
    bit-identical; the weight gradients are summed over a different number of f32 slabs -- f32 rounding, nothing more."""
    from theia_amd import ops
    from theia_amd.parallel import TheiaDataParallel
    bb, teachers, B = "facebook/deit-base-patch16-224", O.TEACHER_SETS["cddsv"], 64
    model, _ = build(bb, teachers, "bf16")
    images = O.synth_images(B, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}

    def step():
        model.zero_grad(set_to_none=True)
        losses = model.get_loss(model(images), targets)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
        return losses

    la = step()
    ga = {k: p.grad.clone() for k, p in model.named_parameters()}
    ddp = TheiaDataParallel(model, broadcast=False)  # world size 1: no collective, the reducer is a no-op
    ddp._reserve = 16
    model.engine.bucket_ready_hook = ddp._on_bucket
    full = ops.device_cus()  # (first use probes the device through set_compute_cus: before the spy goes in)
    calls = []
    real = ops.set_compute_cus
    monkeypatch.setattr(ops, "set_compute_cus", lambda n: (calls.append(n), real(n))[1])
    try:
        lb = step()
    finally:
        model.engine.bucket_ready_hook = None
        real(0)
    assert calls == [full - 16, full] and ops.get_compute_cus() == full, calls  # restored to what it was, not to "0"
    for k in ("mse_loss", "cos_loss", "l1_loss"):
        assert float(la[k]) == float(lb[k])
    worst = 0.0
    for k, p in model.named_parameters():
        a, b = ga[k].double(), p.grad.double()
        worst = max(worst, float((a - b).norm() / a.norm().clamp_min(1e-30)))
    print(f"[base bf16 B=64, 16 CUs reserved during backward] worst |dW - dW'| / |dW| = {worst:.2e}")
    assert worst < 1e-5, worst


def test_input_layouts_and_reduce_modes(golden_dir):
    """G6/G7: BHWC == BCHW == list-of-PIL inputs; handle_feature_output modes (models/utils.py:8-43)."""
    from PIL import Image
    g = np.load(os.path.join(golden_dir, "g6_g7_tokens_layouts.npz"))
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    model, _ = build(bb, teachers, "fp32")
    model.eval()
    images = O.synth_images(2, seed=3)
    with torch.no_grad():
        a = model.forward_feature(images)
        b = model.forward_feature(images.permute(0, 3, 1, 2).contiguous())
        c = model.forward_feature([Image.fromarray(images[i].numpy()) for i in range(2)])
        assert torch.equal(a, b) and torch.equal(a, c)  # the reference is bit-identical across layouts too (G7)
        z = model.backbone(images)
        zz = z.float().cpu().numpy().reshape(-1)
        assert np.abs(zz[g["z_idx"]] - g["z_val"]).max() / np.abs(g["z_val"]).max() < 1e-4
        for mode in ("mean_pooling", "max_pooling", "cls", "identity", None):
            model.feature_reduce_method = mode
            y = model.forward_feature(images)
            key = "none" if mode is None else mode
            assert tuple(y.shape) == tuple(g[f"hfo_{key}_shape"])
            yy = y.float().cpu().numpy().reshape(-1)
            assert np.abs(yy[g[f"hfo_{key}_idx"]] - g[f"hfo_{key}_val"]).max() / np.abs(g[f"hfo_{key}_val"]).max() < 1e-4
        model.feature_reduce_method = "bogus"
        with pytest.raises(NotImplementedError):
            model.forward_feature(images)


def test_non_224_inputs_go_through_the_processor_resize(golden_dir):
    """forward_feature on non-224 images (do_resize=True): against the REFERENCE's forward_feature on the same inputs (golden
    G12: its processor resizes with Pillow), for both layouts, single-pass cases and a list of differently sized images."""
    from PIL import Image
    g = np.load(os.path.join(golden_dir, "g12_resize_processor.npz"))
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    m, _ = build(bb, teachers, "fp32")
    for name in ("up_hwc", "down_chw", "xonly_hwc", "yonly_chw"):
        x = torch.from_numpy(g[f"{name}_img"])
        with torch.no_grad():
            z = m.forward_feature(x).float().cpu().numpy()
        assert rel(np.abs(z.astype(np.float64)).sum(), float(g[f"{name}_z_abssum"])) < 1e-4, name
        assert np.allclose(z.reshape(-1)[g[f"{name}_z_idx"]], g[f"{name}_z_val"], rtol=1e-3, atol=2e-4), name
    # a list of PIL images of different sizes == each image on its own
    a = g["up_hwc_img"][0]
    b_ = np.transpose(g["down_chw_img"][0], (1, 2, 0))
    with torch.no_grad():
        zl = m.forward_feature([Image.fromarray(a), Image.fromarray(np.ascontiguousarray(b_))])
        z0 = m.forward_feature(torch.from_numpy(a[None]))
        z1 = m.forward_feature(torch.from_numpy(g["down_chw_img"]))
    assert torch.equal(zl[0], z0[0]) and torch.equal(zl[1], z1[0])
    with pytest.raises(ValueError, match="doesn't match model"):  # HF ViTEmbeddings raises ValueError for a size mismatch too
        m.forward_feature(torch.from_numpy(a[None]), do_resize=False)


def test_deit_hub_processor_configuration(golden_dir):
    """processor="deit": resize 256 bicubic + center-crop 224 (the hub checkpoints' preprocessor_config) -- forward_feature
    against the reference model run with transformers' DeiTImageProcessor (golden G14), 224 and non-224 inputs."""
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.foundation_models.common import get_model_feature_size
    g = np.load(os.path.join(golden_dir, "g14_deit_processor.npz"))
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    m = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers},
                      precision="fp32", processor="deit")
    m.load_state_dict(O.synth_params(bb, teachers, 0), strict=True)
    m = m.to("cuda:0")
    assert m.backbone.resize_size == (256, 256) and m.backbone.crop_size == 224 and m.backbone.resample == 3
    for name, img in (("in224", O.synth_images(2, 0)), ("in300x260", torch.from_numpy(g["in300x260_img"]))):
        with torch.no_grad():
            z = m.forward_feature(img).float().cpu().numpy()
        assert rel(np.abs(z.astype(np.float64)).sum(), float(g[f"{name}_z_abssum"])) < 1e-4, name
        assert np.allclose(z.reshape(-1)[g[f"{name}_z_idx"]], g[f"{name}_z_val"], rtol=1e-3, atol=2e-4), name
    # back to the default processor: 224x224 input is an identity again
    m.backbone.set_processor()
    with torch.no_grad():
        z0 = m.forward_feature(O.synth_images(2, 0))
    g1 = np.load(os.path.join(golden_dir, "g6_g7_tokens_layouts.npz"))
    assert z0.shape[1] == 196


def test_grad_accumulation_and_freeze_translator():
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    model, _ = build(bb, teachers, "fp32")
    images = O.synth_images(2, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(2, teachers, 1).items()}

    def step():
        losses = model.get_loss(model(images), targets, as_float=False)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()

    step()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    step()  # second backward without zero_grad accumulates
    for k, p in model.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-8), k
    model.zero_grad(set_to_none=True)
    model.freeze_translator()
    step()
    for k, p in model.named_parameters():
        if k.startswith("translator."):
            assert p.grad is None, k
        elif "k_proj.bias" not in k:
            assert torch.allclose(p.grad, g1[k], rtol=1e-5, atol=1e-8), k


def test_state_dict_keys_match_reference_layout(golden_dir):
    g = np.load(os.path.join(golden_dir, CASES["g3"]))
    bb, teachers = str(g["meta_backbone"]), [str(t) for t in g["meta_teachers"]]
    model, _ = build(bb, teachers, "fp32")
    assert sorted(model.state_dict().keys()) == sorted(str(n) for n in g["grad_names"])


def test_forward_feature_is_differentiable_for_every_reduce_mode():
    """fine-tuning the backbone through forward_feature (reference: slices / mean / amax of last_hidden_state are part of
    the autograd graph): the backbone gradients must equal those obtained by reducing the token matrix with torch ops."""
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    model, _ = build(bb, teachers, "fp32")
    images = O.synth_images(2, 3)
    probe = "backbone.model.layers.3.mlp.fc1.weight"
    for mode in (None, "mean_pooling", "max_pooling", "cls"):
        model.feature_reduce_method = mode
        model.zero_grad(set_to_none=True)
        y = model.forward_feature(images)
        assert y.requires_grad and y.dtype == torch.float32
        w = torch.from_numpy(O._hash_uniform(y.numel(), 77).reshape(tuple(y.shape)).copy()).to(y.device)
        (y * w).sum().backward()
        got = dict(model.named_parameters())[probe].grad.clone()
        model.zero_grad(set_to_none=True)
        z = model.backbone(images).float()
        tok = z[:, 1:]
        ref = {None: tok, "mean_pooling": tok.mean(1), "max_pooling": tok.amax(1), "cls": z[:, 0]}[mode]
        assert torch.equal(ref.detach(), y.detach()) or torch.allclose(ref.detach(), y.detach(), rtol=1e-6, atol=1e-7)
        (ref * w).sum().backward()
        want = dict(model.named_parameters())[probe].grad
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-7), mode


@pytest.mark.parametrize("key", ["g15", "g16"])
def test_nocls_and_reg_students_match_reference_goldens(golden_dir, key):
    """The student without CLS token (196 tokens, backbones.py:344-416) and the one with 7 register tokens (204 tokens,
    :419-503) against the reference's own classes (goldens G15 / G16: features, predictions, losses, every gradient norm).
    forward_feature of the nocls- student returns 195 tokens: the reference slices x[:, 1:] whatever the backbone."""
    name = {"g15": "g15_nocls_tiny_dinov2_b2.npz", "g16": "g16_reg_tiny_dinov2_b2.npz"}[key]
    g = np.load(os.path.join(golden_dir, name))
    bb, teachers, B = str(g["meta_backbone"]), [str(t) for t in g["meta_teachers"]], int(g["meta_B"])
    model, _ = build(bb, teachers, "fp32")
    assert model.no_cls == (key == "g15") and model.num_reg_tokens == (7 if key == "g16" else 0)
    assert sorted(model.state_dict().keys()) == sorted(str(n) for n in g["grad_names"])
    images = O.synth_images(B, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}
    with torch.no_grad():
        feat = model.forward_feature(images)
    assert tuple(feat.shape) == tuple(g["feat_shape"])
    f = feat.float().cpu().numpy().reshape(-1)
    assert np.abs(f[g["feat_idx"]] - g["feat_val"]).max() / np.abs(g["feat_val"]).max() < 1e-4
    assert rel(np.abs(f.astype(np.float64)).sum(), float(g["feat_abssum"])) < 1e-5
    pred = model(images)
    for ti, t in enumerate(teachers):
        p = pred[t].detach().float().cpu().numpy().reshape(-1)
        assert np.abs(p[g[f"pred{ti}_idx"]] - g[f"pred{ti}_val"]).max() / np.abs(g[f"pred{ti}_val"]).max() < 1e-4
    losses = model.get_loss(pred, targets)
    for k in ("mse_loss", "cos_loss", "l1_loss"):
        assert rel(float(losses[k]), float(g[k])) < 1e-4, k
    (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    sdp = dict(model.named_parameters())
    for i, k in enumerate(str(n) for n in g["grad_names"]):
        gn = float(sdp[k].grad.double().pow(2).sum().sqrt())
        if "k_proj.bias" in k:
            assert gn < 1e-6 * max(1.0, g["gradnorm_cos_l1"].max()), k
            continue
        assert rel(gn, g["gradnorm_cos_l1"][i]) < 1e-3, (k, gn, g["gradnorm_cos_l1"][i])
    # the throughput mode on the same students
    mb, params = build(bb, teachers, "bf16")
    lb = mb.get_loss(mb(images), targets)
    for k in ("mse_loss", "cos_loss", "l1_loss"):
        assert rel(float(lb[k]), float(g[k])) < 2e-2, k


@pytest.mark.parametrize("tag,bb", [("deit", "facebook/deit-tiny-patch16-224"), ("nocls", "nocls-facebook/deit-tiny-patch16-224"),
                                    ("reg", "reg-facebook/deit-tiny-patch16-224")])
def test_interpolated_position_embeddings_match_reference(golden_dir, tag, bb):
    """160x192 images with do_resize=False, interpolate_pos_encoding=True (120 patch tokens) against the reference (golden
    G17): features, and the gradients through the bicubic interpolation into position_embeddings, a mid-depth weight and the
    patch projection.  DeiT uses HF's size-based interpolation, nocls- / reg- the reference's own scale_factor variant."""
    g = np.load(os.path.join(golden_dir, "g17_interpolate_pos.npz"))
    model, _ = build(bb, O.TEACHER_SETS["dinov2"], "fp32")
    img = torch.from_numpy(g["img"])
    with pytest.raises(ValueError, match="doesn't match model"):
        model.forward_feature(img, do_resize=False)
    z = model.forward_feature(img, do_resize=False, interpolate_pos_encoding=True)
    assert tuple(z.shape) == tuple(g[f"{tag}_z_shape"])
    zz = z.detach().float().cpu().numpy().reshape(-1)
    assert np.abs(zz[g[f"{tag}_z_idx"]] - g[f"{tag}_z_val"]).max() / np.abs(g[f"{tag}_z_val"]).max() < 1e-4
    assert rel(np.abs(zz.astype(np.float64)).sum(), float(g[f"{tag}_z_abssum"])) < 1e-5
    w = torch.from_numpy(O._hash_uniform(z.numel(), 77).reshape(tuple(z.shape)).copy()).to(z.device)
    (z * w).sum().backward()
    sdp = dict(model.named_parameters())
    for short, k in (("pos", "backbone.model.embeddings.position_embeddings"), ("fc1", "backbone.model.layers.3.mlp.fc1.weight"),
                     ("patch", "backbone.model.embeddings.patch_embeddings.projection.weight")):
        gr = sdp[k].grad.detach().float().cpu().numpy().reshape(-1)
        assert rel(float(np.sqrt((gr.astype(np.float64) ** 2).sum())), float(g[f"{tag}_g{short}_norm"])) < 1e-3, (tag, short)
        assert np.abs(gr[g[f"{tag}_g{short}_idx"]] - g[f"{tag}_g{short}_val"]).max() <= 1e-2 * float(g[f"{tag}_g{short}_norm"]) / np.sqrt(gr.size) + 1e-9
    # same images through the default path (processor resize to 224): 196 patch tokens again
    with torch.no_grad():
        assert model.forward_feature(img).shape[1] == (195 if tag == "nocls" else 196)


@pytest.mark.parametrize("bb", ["facebook/deit-tiny-patch16-224", "facebook/deit-small-patch16-224"])
def test_fp8_mode_tracks_the_oracle(bb):
    """precision="fp8" (BASELINE configs[3]: e4m3 operands with per-tensor delayed scaling for the forward and data-gradient
    GEMMs, bf16 activations / epilogues / weight-gradient GEMMs, f32 accumulation) against the CPU oracle's fp32 losses and
    gradients.  Stated tolerance: losses 5e-2 relative, per-tensor gradient cosine > 0.9 (e4m3 carries 3 mantissa bits: ~6 %
    per operand element, averaged down by the K-long dot products)."""
    teachers, B = O.TEACHER_SETS["cddsv"], 4
    model, params = build(bb, teachers, "fp8")
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, teachers, 1)
    targets = {t: v.to("cuda:0") for t, v in tcpu.items()}
    for _ in range(2):  # second pass: every quantisation site has a calibrated scale and re-uses it
        model.zero_grad(set_to_none=True)
        losses = model.get_loss(model(images), targets)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    assert model.engine.fp8 is not None and len(model.engine.fp8.index) > 100
    ref_losses, _, grads, _ = O.train_step_grads(params, images, tcpu, bb, teachers, "cos_l1")
    for k in ("mse_loss", "cos_loss", "l1_loss"):
        assert rel(float(losses[k]), float(ref_losses[k])) < 5e-2, (k, float(losses[k]), float(ref_losses[k]))
    cos, nr, who = _grad_agreement(model, grads)
    print(f"[fp8 {bb}] worst gradient cosine {cos:.4f} ({who}), worst norm-ratio error {nr:.3f}")
    # (round 6: DeiT-tiny 0.918 -- the 3x3 convolutions and the attention-adjacent projections keep bf16 operands now; rounds 3-5: 0.898-0.91 with
    # a gate of 0.88; DeiT-small / base sit at 0.93+)
    assert cos > 0.9 and nr < 0.15, (cos, nr, who)


@pytest.mark.parametrize("key", ["g1", "g2"])
def test_hub_loaded_model_matches_reference_goldens(golden_dir, key, tmp_path):
    """SURVEY f4 (reference README.md:22-38, models/rvfm.py:77-87): a hub-layout snapshot (config.json + model.safetensors, and a
    transformers-4.4x-era pytorch_model.bin with foreign keys) loaded through ``TheiaModel.from_pretrained`` / ``AutoModel`` runs on
    the GPU and reproduces the reference's goldens G1 (DeiT-tiny + dinov2, B = 8) and G2 (tiny + cdiv, B = 2) end to end."""
    import json
    from theia_amd.hub import TheiaModel, register_with_transformers
    g, bb, teachers, B = load_case(golden_dir, key)
    src, _ = build(bb, teachers, "fp32")
    hub = TheiaModel(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                     target_feature_sizes=src.target_feature_sizes, precision="fp32")
    hub.load_state_dict(src.state_dict(), strict=True)
    snap = str(tmp_path / "snap")
    hub.save_pretrained(snap)
    del hub, src
    if key == "g1":  # safetensors snapshot through AutoModel
        register_with_transformers()
        from transformers import AutoModel
        model = AutoModel.from_pretrained(snap, device="cuda:0", precision="fp32")
    else:  # legacy key names + keys the model does not have, as a .bin pickle (weights_only load)
        from safetensors.torch import load_file
        legacy = {}
        for k, v in load_file(os.path.join(snap, "model.safetensors")).items():
            k2 = (k.replace(".layers.", ".encoder.layer.").replace(".attention.q_proj.", ".attention.attention.query.")
                   .replace(".attention.k_proj.", ".attention.attention.key.").replace(".attention.v_proj.", ".attention.attention.value.")
                   .replace(".attention.o_proj.", ".attention.output.dense.").replace(".mlp.fc1.", ".intermediate.dense.")
                   .replace(".mlp.fc2.", ".output.dense."))
            legacy[k2] = v
        legacy["backbone.model.pooler.dense.weight"] = torch.zeros(4, 4)
        old = tmp_path / "old"
        os.makedirs(old)
        torch.save(legacy, old / "pytorch_model.bin")
        json.dump({"backbone": bb, "target_model_names": teachers, "precision": "fp32"}, open(old / "config.json", "w"))
        model = TheiaModel.from_pretrained(str(old), device="cuda:0")
        assert model.loading_info["unexpected_keys"] == 1
    assert isinstance(model, TheiaModel) and model.loading_info["missing_keys"] == [] and next(model.parameters()).is_cuda
    check_against_golden(model, g, teachers, B, key + "/hub")


def test_hub_loader_refuses_a_snapshot_that_matches_nothing(tmp_path):
    import json
    from theia_amd.hub import TheiaModel
    os.makedirs(tmp_path / "bad")
    torch.save({"module.encoder.w": torch.zeros(3)}, tmp_path / "bad" / "pytorch_model.bin")
    json.dump({"backbone": "facebook/deit-tiny-patch16-224", "target_model_names": O.TEACHER_SETS["dinov2"]}, open(tmp_path / "bad" / "config.json", "w"))
    with pytest.raises(RuntimeError, match="none of its"):
        TheiaModel.from_pretrained(str(tmp_path / "bad"))


def test_bf16_training_tracks_the_fp32_oracle_trajectory():
    """20 optimisation steps of DeiT-tiny + dinov2 at B = 8 (BASELINE configs[0]'s model) on one fixed batch, lr 2e-4 (at 1e-3 this
    over-fitting run is unstable -- loss spikes at step 8 -- and even the fp32 engine and the fp32 oracle part ways after ~13 steps):
    the CPU oracle in fp32 with a plain AdamW written out here (decay rule of optimizers/utils.py:8-35) against the engine with
    FusedAdamW in fp32 and in bf16.  Adam's normalisation amplifies last-bit gradient differences, so even the fp32 engine drifts
    from the fp32 oracle to 1.7e-3 of the loss by step 20 (first steps: 1e-7); bf16 ends at 3.4e-3 -- twice the fp32 figure, while the
    loss itself falls by 13 %.  Gates: fp32 < 5e-3, bf16 < 1e-2 at every step, the first step < 1e-6 / 1e-4, and the same total
    decrease within 5 %."""
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.optimizers.utils import is_no_decay
    bb, teachers, B, steps = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"], 8, 20
    lr, b1, b2, eps, wd = 2e-4, 0.9, 0.999, 1e-8, 0.01
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, teachers, 1)
    targets = {t: v.to("cuda:0") for t, v in tcpu.items()}
    curves = {}
    for prec in ("fp32", "bf16"):
        model, params = build(bb, teachers, prec)
        opt = FusedAdamW(model, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
        got = []
        for _ in range(steps):
            opt.zero_grad()
            losses = model.get_loss(model(images), targets, as_float=False)
            main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
            main.backward()
            opt.step()
            got.append(float(main))
        curves[prec] = got
        del model, opt
    ref = []
    P = {k: v.clone() for k, v in params.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    for t in range(1, steps + 1):
        _l, main, grads, _p = O.train_step_grads(P, images, tcpu, bb, teachers, "cos_l1")
        ref.append(float(main))
        for k in P:
            gk = grads[k]
            decay = 0.0 if is_no_decay(k, P[k]) else wd
            P[k].mul_(1.0 - lr * decay)
            M[k].mul_(b1).add_(gk, alpha=1.0 - b1)
            V[k].mul_(b2).addcmul_(gk, gk, value=1.0 - b2)
            denom = (V[k] / (1.0 - b2 ** t)).sqrt_().add_(eps)
            P[k].addcdiv_(M[k] / (1.0 - b1 ** t), denom, value=-lr)
    dev32 = max(abs(a - b) / abs(b) for a, b in zip(curves["fp32"], ref))
    dev16 = max(abs(a - b) / abs(b) for a, b in zip(curves["bf16"], ref))
    print(f"[trajectory] oracle {ref[0]:.5f} -> {ref[-1]:.5f}; fp32 engine worst step deviation {dev32:.2e}; "
          f"bf16 {curves['bf16'][0]:.5f} -> {curves['bf16'][-1]:.5f}, worst step deviation {dev16:.2e}")
    drop = ref[0] - ref[-1]
    assert drop > 0.02 and all(b < a for a, b in zip(ref, ref[1:]))  # a smooth, optimising run
    assert dev32 < 5e-3 and abs(curves["fp32"][0] - ref[0]) < 1e-6 * ref[0], (curves["fp32"], ref)
    assert dev16 < 1e-2 and abs(curves["bf16"][0] - ref[0]) < 1e-4 * ref[0], (curves["bf16"], ref)
    assert abs((curves["bf16"][0] - curves["bf16"][-1]) - drop) < 5e-2 * drop


def test_bf16_training_of_the_headline_model_tracks_fp32_over_20_steps():
    """Review item 9 (round 4): the trajectory test above runs DeiT-tiny; this one runs the BENCHED model -- DeiT-base + the five cddsv
    teachers, every GEMM on the library's dispatch -- for 20 optimizer steps at B = 8 on one fixed batch, in bf16 and in fp32.  The fp32
    engine is the reference here (a CPU-oracle step of this model takes ~35 s; the fp32 engine itself is held to the reference goldens
    G3 / G5 and to the oracle at 1e-4 by the tests above, and to the oracle's TRAJECTORY by the DeiT-tiny test).  Gates: every bf16 step
    within 1e-2 of the fp32 loss, the first step within 1e-3 (same weights: pure forward rounding), the same total decrease within 5 %,
    and the three loss terms individually within 2e-2 at the last step."""
    from theia_amd.optimizers import FusedAdamW
    bb, teachers, B, steps = "facebook/deit-base-patch16-224", O.TEACHER_SETS["cddsv"], 8, 20
    images = O.synth_images(B, 0)
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}
    curves, last = {}, {}
    for prec in ("fp32", "bf16"):
        model, _params = build(bb, teachers, prec)
        opt = FusedAdamW(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        got = []
        for _ in range(steps):
            opt.zero_grad()
            losses = model.get_loss(model(images), targets, as_float=False)
            main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
            main.backward()
            opt.step()
            got.append(float(main))
        curves[prec] = got
        last[prec] = {k: float(losses[k]) for k in ("mse_loss", "cos_loss", "l1_loss")}
        del model, opt
        torch.cuda.empty_cache()
    f32, b16 = curves["fp32"], curves["bf16"]
    dev = [abs(a - b) / abs(b) for a, b in zip(b16, f32)]
    print("headline-model trajectory: fp32", [round(v, 5) for v in f32[::5]], "bf16", [round(v, 5) for v in b16[::5]], "max rel dev", max(dev))
    assert all(v == v for v in b16) and f32[-1] < f32[0]
    assert dev[0] < 1e-3 and max(dev) < 1e-2, dev
    assert abs((b16[0] - b16[-1]) - (f32[0] - f32[-1])) < 0.05 * abs(f32[0] - f32[-1])
    for k in last["fp32"]:
        assert abs(last["bf16"][k] - last["fp32"][k]) < 2e-2 * abs(last["fp32"][k]), (k, last)


def test_headline_model_training_tracks_the_oracle_over_three_steps():
    """Round-5 review item 7: the 20-step headline test above holds the bf16 engine to the fp32 ENGINE; this one holds both to the
    ORACLE on the benched model -- DeiT-base + the five cddsv teachers, library dispatch, B = 2, three AdamW steps (an oracle step of this
    model is ~10 s of host time; lr 1e-4, decay rule of optimizers/utils.py:8-35 written out as in the DeiT-tiny trajectory test).  After
    the first update the parameters of engine and oracle differ by what Adam's normalisation makes of last-bit gradient differences, so the
    later losses test the optimizer path too.  Gates: fp32 engine within 1e-4 of the oracle's loss at step 1 and 2e-3 at every step, bf16
    within 2e-3 at step 1 and 1e-2 at every step, and all three decrease."""
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.optimizers.utils import is_no_decay
    bb, teachers, B, steps = "facebook/deit-base-patch16-224", O.TEACHER_SETS["cddsv"], 2, 3
    lr, b1, b2, eps, wd = 1e-4, 0.9, 0.999, 1e-8, 0.01
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, teachers, 1)
    targets = {t: v.to("cuda:0") for t, v in tcpu.items()}
    curves = {}
    for prec in ("fp32", "bf16"):
        model, params = build(bb, teachers, prec)
        opt = FusedAdamW(model, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
        got = []
        for _ in range(steps):
            opt.zero_grad()
            losses = model.get_loss(model(images), targets, as_float=False)
            main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
            main.backward()
            opt.step()
            got.append(float(main))
        curves[prec] = got
        del model, opt
        torch.cuda.empty_cache()
    ref = []
    P = {k: v.clone() for k, v in params.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    for t in range(1, steps + 1):
        _l, main, grads, _p = O.train_step_grads(P, images, tcpu, bb, teachers, "cos_l1")
        ref.append(float(main))
        if t == steps:
            break
        for k in P:
            gk = grads[k]
            decay = 0.0 if is_no_decay(k, P[k]) else wd
            P[k].mul_(1.0 - lr * decay)
            M[k].mul_(b1).add_(gk, alpha=1.0 - b1)
            V[k].mul_(b2).addcmul_(gk, gk, value=1.0 - b2)
            denom = (V[k] / (1.0 - b2 ** t)).sqrt_().add_(eps)
            P[k].addcdiv_(M[k] / (1.0 - b1 ** t), denom, value=-lr)
    d32 = [abs(a - b) / abs(b) for a, b in zip(curves["fp32"], ref)]
    d16 = [abs(a - b) / abs(b) for a, b in zip(curves["bf16"], ref)]
    print(f"[headline trajectory vs oracle] oracle {[round(v, 6) for v in ref]}; fp32 engine deviations {[f'{v:.1e}' for v in d32]}; "
          f"bf16 {[f'{v:.1e}' for v in d16]}")
    assert all(b < a for a, b in zip(ref, ref[1:])) and all(b < a for a, b in zip(curves["bf16"], curves["bf16"][1:]))
    assert d32[0] < 1e-4 and max(d32) < 2e-3, (curves["fp32"], ref)
    assert d16[0] < 2e-3 and max(d16) < 1e-2, (curves["bf16"], ref)


def test_fp8_training_tracks_the_fp32_oracle_trajectory_on_deit_small():
    """BASELINE configs[3]'s student against the ORACLE over optimizer steps (round-5 review item 4): DeiT-small + cdiv, B = 4, 20 AdamW steps
    at lr 2e-4 on one fixed batch -- the CPU oracle in fp32 with the written-out AdamW of the trajectory tests above, the engine in fp8
    mode (e4m3 operands with delayed per-tensor scales refreshed every 32nd step, quantisation fused into the producers, bf16
    weight-gradient / attention-adjacent / 3x3-kernel launches) and, for scale, in bf16.  Gates: fp8 within 2e-2 of the oracle's loss at
    every step and 5e-3 at the first, the same total decrease within 10 %; bf16 under the same 2e-2 (it measures 1.2e-2 here: on this
    over-fitting run of 4 images the loss curve has a kink at step 6 where any rounding realisation lands slightly differently)."""
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.optimizers.utils import is_no_decay
    bb, teachers, B, steps = "facebook/deit-small-patch16-224", O.TEACHER_SETS["cdiv"], 4, 20
    lr, b1, b2, eps, wd = 2e-4, 0.9, 0.999, 1e-8, 0.01
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, teachers, 1)
    targets = {t: v.to("cuda:0") for t, v in tcpu.items()}
    curves = {}
    for prec in ("bf16", "fp8"):
        model, params = build(bb, teachers, prec)
        opt = FusedAdamW(model, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
        got = []
        for _ in range(steps):
            opt.zero_grad()
            losses = model.get_loss(model(images), targets, as_float=False)
            main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
            main.backward()
            opt.step()
            got.append(float(main))
        curves[prec] = got
        del model, opt
        torch.cuda.empty_cache()
    ref = []
    P = {k: v.clone() for k, v in params.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    for t in range(1, steps + 1):
        _l, main, grads, _p = O.train_step_grads(P, images, tcpu, bb, teachers, "cos_l1")
        ref.append(float(main))
        for k in P:
            gk = grads[k]
            decay = 0.0 if is_no_decay(k, P[k]) else wd
            P[k].mul_(1.0 - lr * decay)
            M[k].mul_(b1).add_(gk, alpha=1.0 - b1)
            V[k].mul_(b2).addcmul_(gk, gk, value=1.0 - b2)
            denom = (V[k] / (1.0 - b2 ** t)).sqrt_().add_(eps)
            P[k].addcdiv_(M[k] / (1.0 - b1 ** t), denom, value=-lr)
    d8 = [abs(a - b) / abs(b) for a, b in zip(curves["fp8"], ref)]
    d16 = [abs(a - b) / abs(b) for a, b in zip(curves["bf16"], ref)]
    print(f"[fp8 trajectory, DeiT-small] oracle {ref[0]:.5f} -> {ref[-1]:.5f}; fp8 {curves['fp8'][0]:.5f} -> {curves['fp8'][-1]:.5f}, worst step "
          f"deviation {max(d8):.2e} (first {d8[0]:.2e}); bf16 worst {max(d16):.2e}")
    drop = ref[0] - ref[-1]
    assert drop > 0.01 and ref[-1] < ref[0]
    assert d8[0] < 5e-3 and max(d8) < 2e-2, (curves["fp8"], ref)
    assert max(d16) < 2e-2, (curves["bf16"], ref)
    assert abs((curves["fp8"][0] - curves["fp8"][-1]) - drop) < 0.10 * drop
