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
