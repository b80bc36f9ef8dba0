"""cProfile of the host side of one eager train step (what the Python layer spends per step while the GPU is busy).
    python tools/host_profile.py [--backbone ...] [--teachers cdiv] [--batch 256]"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from theia_amd.foundation_models.common import get_model_feature_size  # noqa: E402
from theia_amd.models.rvfm import RobotVisionFM  # noqa: E402
from theia_amd.optimizers import FusedAdamW  # noqa: E402

TEACHER_SETS = {"dinov2": ["facebook/dinov2-large"], "cdiv": ["google/vit-huge-patch14-224-in21k", "facebook/dinov2-large", "openai/clip-vit-large-patch14"],
                "cddsv": ["google/vit-huge-patch14-224-in21k", "facebook/dinov2-large", "openai/clip-vit-large-patch14", "facebook/sam-vit-huge",
                          "LiheYoung/depth-anything-large-hf"]}
ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="facebook/deit-tiny-patch16-224")
ap.add_argument("--teachers", default="cdiv")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
T = TEACHER_SETS[a.teachers]
model = RobotVisionFM(backbone=a.backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in T}, precision="bf16").to(dev)
opt = FusedAdamW(model, lr=1e-4)
images = torch.randint(0, 256, (a.batch, 224, 224, 3), dtype=torch.uint8).to(dev)
targets = {t: torch.randn(a.batch, get_model_feature_size(t, True)[1] * get_model_feature_size(t, True)[2], get_model_feature_size(t, True)[0]).to(dev) for t in T}


def step():
    opt.zero_grad(set_to_none=True)
    losses = model.get_loss(model(images), targets, as_float=False)
    (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
# the backward functions run in the autograd engine's own (C++-created) thread, which a profiler enabled in the main thread does not see:
# each custom backward switches its own profiler on around its body
from theia_amd import engine as E  # noqa: E402

bw = cProfile.Profile()


def wrap(cls):
    orig = cls.backward

    def backward(ctx, *g):
        bw.enable()
        try:
            return orig(ctx, *g)
        finally:
            bw.disable()
    cls.backward = staticmethod(backward)


for name in ("_BackboneFn", "_TranslatorFn", "_LossFn"):
    if hasattr(E, name):
        wrap(getattr(E, name))
pr = cProfile.Profile()
import time  # noqa: E402
t0 = time.perf_counter()
pr.enable()
for _ in range(a.steps):
    step()
pr.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host time per step (profiled): {host / a.steps * 1e3:.2f} ms")
print("=== main thread (forward, optimizer)")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
print("=== autograd thread (the custom backward functions)")
pstats.Stats(bw).sort_stats("tottime").print_stats(28)
