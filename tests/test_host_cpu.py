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
