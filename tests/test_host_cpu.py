"""CPU: host-side pieces that need no GPU -- configuration composition (reference YAML tree, scientific notation),
bench.py's multi-GPU launcher path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONFIGS = "/root/reference/src/theia/configs"


def test_compose_parses_scientific_notation_overrides():
    from theia_amd.utils import config as c
    cfg = c.compose(["training.base_lr=2e-3", "training.weight_decay=1E-2", "training.lr_scheduler.warm_up_lr_start_factor=5e-3",
                     "logging.notes=1e3x", "training.batch_size=128"])
    assert cfg.training.base_lr == 2e-3 and isinstance(cfg.training.base_lr, float)
    assert cfg.training.weight_decay == 1e-2 and cfg.training.lr_scheduler.warm_up_lr_start_factor == 5e-3
    assert cfg.logging.notes == "1e3x" and cfg.training.batch_size == 128  # not everything with an 'e' is a number
    lr = cfg.training.base_lr * (cfg.training.batch_size * 8) / (cfg.training.base_batch_size * cfg.training.base_world_size)
    assert abs(lr - 4e-3) < 1e-12


def test_coerce_is_recursive_and_conservative():
    from theia_amd.utils.config import _coerce
    assert _coerce({"a": "2e-3", "b": ["1e-2", "x", 3], "c": {"d": "-.5E+1"}}) == {"a": 2e-3, "b": [1e-2, "x", 3], "c": {"d": -5.0}}
    assert _coerce("e3") == "e3" and _coerce("1e") == "1e" and _coerce("0x1e3") == "0x1e3"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="the reference tree only exists in the build container")
def test_compose_from_the_reference_yaml_tree():
    """--config-path onto the reference's own configs: same groups / keys, floats where the YAML says 2e-3 / 1e-2."""
    from theia_amd.utils import config as c
    cfg = c.compose(["training/target_models=cddsv", "model.backbone.backbone=facebook/deit-base-patch16-224"], config_path=REF_CONFIGS)
    assert isinstance(cfg.training.base_lr, float) and cfg.training.base_lr == 2e-3
    assert isinstance(cfg.training.lr_scheduler.warm_up_lr_start_factor, float)
    assert len(cfg.training.target_models.target_model_names) == 5
    assert cfg.model.backbone.backbone == "facebook/deit-base-patch16-224" and cfg.model.translator.type == "lconv"
    builtin = c.compose(["training/target_models=cddsv"])
    for k in ("base_lr", "batch_size", "weight_decay", "warm_up_steps_ratio", "main_loss", "epochs"):
        assert cfg.training[k] == builtin.training[k], k


def test_bench_multi_gpu_request_enters_the_launcher():
    """`bench.py --gpus 2` without a torch.distributed environment becomes the launcher; with fewer GPUs than ranks it
    reports that in a JSON line (here: 0 GPUs) instead of asking the caller to use torchrun."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 2, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and "GPU" in line["error"]


def test_hub_style_loader_round_trip_and_legacy_keys(tmp_path):
    """theia_amd.hub.TheiaModel: save_pretrained -> from_pretrained on a hub-layout directory (config.json + model.safetensors),
    a checkpoint with transformers-4.4x-era key names + foreign keys (weight filter of rvfm.py:77-87), and AutoModel dispatch."""
    import torch
    from oracle import theia_oracle as O
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.hub import TheiaModel, register_with_transformers
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["cdiv"]
    m = TheiaModel(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0}, feature_reduce_method="cls",
                   target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers})
    m.load_state_dict(O.synth_params(bb, teachers, 0), strict=True)
    m.save_pretrained(str(tmp_path / "snap"))
    assert sorted(os.listdir(tmp_path / "snap")) == ["config.json", "model.safetensors"]
    m2 = TheiaModel.from_pretrained(str(tmp_path / "snap"))
    assert m2.feature_reduce_method == "cls" and list(m2.target_feature_sizes) == teachers and m2.loading_info["missing_keys"] == []
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # keyword overrides like AutoModel.from_pretrained(..., feature_reduce_method=...)
    assert TheiaModel.from_pretrained(str(tmp_path / "snap"), feature_reduce_method=None).feature_reduce_method is None
    # 4.4x-era names + keys the model does not have
    legacy = {}
    for k, v in m.state_dict().items():
        k2 = (k.replace(".layers.", ".encoder.layer.").replace(".attention.q_proj.", ".attention.attention.query.")
               .replace(".attention.k_proj.", ".attention.attention.key.").replace(".attention.v_proj.", ".attention.attention.value.")
               .replace(".attention.o_proj.", ".attention.output.dense.").replace(".mlp.fc1.", ".intermediate.dense.")
               .replace(".mlp.fc2.", ".output.dense."))
        legacy[k2] = v.clone()
    legacy["backbone.model.pooler.dense.weight"] = torch.zeros(4, 4)
    os.makedirs(tmp_path / "old")
    torch.save(legacy, tmp_path / "old" / "pytorch_model.bin")
    json.dump({"backbone": bb, "target_model_names": teachers}, open(tmp_path / "old" / "config.json", "w"))
    m3 = TheiaModel.from_pretrained(str(tmp_path / "old"))
    assert m3.loading_info["missing_keys"] == [] and m3.loading_info["unexpected_keys"] == 1
    for k, v in m.state_dict().items():
        assert torch.equal(v, m3.state_dict()[k]), k
    register_with_transformers()
    from transformers import AutoModel
    m4 = AutoModel.from_pretrained(str(tmp_path / "snap"))
    assert isinstance(m4, TheiaModel) and torch.equal(m4.state_dict()["backbone.model.layernorm.weight"], m.state_dict()["backbone.model.layernorm.weight"])


def test_backbone_variants_build_with_reference_parameter_names():
    from oracle import theia_oracle as O
    from theia_amd.models.rvfm import RobotVisionFM
    for name, ntok in (("nocls-facebook/deit-tiny-patch16-224", 196), ("reg-facebook/deit-tiny-patch16-224", 204), ("facebook/deit-tiny-patch16-224", 197)):
        m = RobotVisionFM(backbone=name, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0}, target_feature_sizes=None)
        assert sorted(k for k in m.state_dict()) == sorted(k for k in O.param_shapes(name, []))
        for k, shp in O.param_shapes(name, []).items():
            assert tuple(m.state_dict()[k].shape) == tuple(shp), k
        assert m.engine.geo224.ntok == ntok and m.no_cls == name.startswith("nocls") and m.num_reg_tokens == (7 if name.startswith("reg") else 0)
    m = RobotVisionFM(backbone="reg-facebook/deit-tiny-patch16-224", translator="lconv", target_feature_sizes=None, num_reg_tokens=4)
    assert m.num_reg_tokens == 4 and m.state_dict()["backbone.model.embeddings.reg_token"].shape == (1, 4, 192)
