"""GPU: the hipGraph-captured, streamed forward_feature (BASELINE configs[4]; theia_amd/streaming.py) against the eager
forward_feature (itself pinned to the reference by goldens G1-G7): bit-identical features for device-resident and
host-resident batches, ragged last chunks, every token-reduce mode, repeated calls, and after a parameter update."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import theia_oracle as O  # noqa: E402  (checker only)


def _model(bb="facebook/deit-tiny-patch16-224", precision="bf16"):
    from theia_amd.models.rvfm import RobotVisionFM
    m = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0}, target_feature_sizes=None,
                      precision=precision)
    sd = {k: v for k, v in O.synth_params(bb, O.TEACHER_SETS["dinov2"], 0).items() if k.startswith("backbone.")}  # as in golden G1
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0").eval()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_streamed_graph_replay_equals_eager(precision):
    m = _model(precision=precision)
    B, chunk = 70, 32  # 2 full chunks + a ragged one
    imgs = O.synth_images(B, 9)
    with torch.no_grad():
        ref = torch.cat([m.forward_feature(imgs[i:i + 16]) for i in range(0, B, 16)])
    got_dev = m.forward_feature_streamed(imgs.to("cuda:0"), chunk=chunk)
    assert got_dev.shape == ref.shape == (B, 196, 192) and got_dev.dtype == ref.dtype
    assert torch.equal(got_dev, ref)
    got_host = m.forward_feature_streamed(imgs.pin_memory(), chunk=chunk)  # H2D staged on the copy stream
    assert torch.equal(got_host, ref)
    sff = m._streamers[(chunk, ())]
    assert sff.replays == 6 and len(sff._graphs) == 1  # one capture, replayed per chunk
    # channels-first layout gets its own capture and the same values
    got_chw = m.forward_feature_streamed(imgs.permute(0, 3, 1, 2).contiguous().to("cuda:0"), chunk=chunk)
    assert torch.equal(got_chw, ref)
    # caller-provided output buffer
    out = torch.empty_like(ref)
    assert m.forward_feature_streamed(imgs.to("cuda:0"), chunk=chunk, out=out) is out and torch.equal(out, ref)


def test_streamed_reduce_modes_and_parameter_updates():
    m = _model()
    imgs = O.synth_images(40, 4).to("cuda:0")
    for mode in ("cls", "mean_pooling", "max_pooling", None):
        m.feature_reduce_method = mode
        with torch.no_grad():
            ref = m.forward_feature(imgs)
        got = m.forward_feature_streamed(imgs, chunk=16)
        assert torch.equal(got, ref), mode
    # a parameter update invalidates the capture (the operand cache is rebuilt outside the graph)
    with torch.no_grad():
        m.backbone.model.layernorm.weight.mul_(1.5)
        ref2 = m.forward_feature(imgs)
    got2 = m.forward_feature_streamed(imgs, chunk=16)
    assert torch.equal(got2, ref2) and not torch.equal(ref2, ref)


def test_streamed_matches_reference_golden(golden_dir):
    """the captured path against the reference's own forward_feature output (golden G1: DeiT-tiny, fp32, B = 8)"""
    import os
    g = np.load(os.path.join(golden_dir, "g1_tiny_dinov2_b8.npz"))
    m = _model(precision="fp32")
    feat = m.forward_feature_streamed(O.synth_images(8, 0).to("cuda:0"), chunk=3)
    f = feat.float().cpu().numpy().reshape(-1)
    assert np.abs(f[g["feat_idx"]] - g["feat_val"]).max() / np.abs(g["feat_val"]).max() < 1e-4


def test_streamed_base_bf16_matches_the_oracle():
    """BASELINE configs[4]'s own model and precision: DeiT-base, bf16, streamed through the captured graph (ragged last chunk) ==
    the eager forward_feature bit for bit, and within bf16 accuracy of the CPU oracle's fp32 features (max-norm 2e-2, cosine > 0.9995)."""
    bb = "facebook/deit-base-patch16-224"
    m = _model(bb, "bf16")
    B, chunk = 40, 16
    imgs = O.synth_images(B, 3)
    with torch.no_grad():  # eager in the streamer's own chunking: the GEMM tile choice (hence the f32 summation order) follows the batch size
        eager = torch.cat([m.forward_feature(imgs[i:i + chunk]) for i in range(0, 32, chunk)] + [m.forward_feature(torch.cat([imgs[32:], imgs[:8]]))[:8]])
    got = m.forward_feature_streamed(imgs.to("cuda:0"), chunk=chunk)
    assert torch.equal(got, eager) and got.shape == (B, 196, 768)
    params = {k: v for k, v in O.synth_params(bb, O.TEACHER_SETS["dinov2"], 0).items() if k.startswith("backbone.")}
    torch.set_num_threads(max(1, min(32, __import__("os").cpu_count() or 1)))
    with torch.no_grad():
        ref = O.forward_feature(params, imgs[:12], bb)
    a, b = got[:12].float().cpu().reshape(-1).double(), ref.reshape(-1).double()
    assert float((a - b).abs().max() / b.abs().max()) < 2e-2
    assert float((a @ b) / (a.norm() * b.norm())) > 0.9995


def test_streamed_call_right_behind_an_optimizer_step():
    """The capture path (warm-up forwards + graph capture on a private stream) must be ordered behind work still queued on the
    caller's stream: an optimizer step is enqueued and the streamed call follows WITHOUT a host synchronisation; the features must be
    those of the updated parameters (a capture that ran ahead would cache operands built from half-updated weights)."""
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]
    m = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers}, precision="bf16")
    m.load_state_dict(O.synth_params(bb, teachers, 0))
    m = m.to("cuda:0")
    opt = FusedAdamW(m, lr=5e-2, weight_decay=0.0)
    imgs = O.synth_images(24, 5).to("cuda:0")
    targets = {t: v.to("cuda:0") for t, v in O.synth_targets(24, teachers, 1).items()}
    for _ in range(3):
        opt.zero_grad()
        losses = m.get_loss(m(imgs), targets, as_float=False)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
        opt.step()
        got = m.forward_feature_streamed(imgs, chunk=8)  # no synchronisation in between
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = m.forward_feature(imgs)
        assert torch.equal(got, ref)
