#!/usr/bin/env python
"""The per-step operand rebuild (theia_cast_batch: every GEMM operand from the fp32 masters, one launch) of the bench model, HIP-event time
per launch.   python tools/cast_bench.py [--backbone facebook/deit-base-patch16-224] [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd.foundation_models.common import get_model_feature_size  # noqa: E402
from theia_amd.models.rvfm import RobotVisionFM  # noqa: E402

T = ["google/vit-huge-patch14-224-in21k", "facebook/dinov2-large", "openai/clip-vit-large-patch14", "facebook/sam-vit-huge", "LiheYoung/depth-anything-large-hf"]
ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="facebook/deit-base-patch16-224")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = RobotVisionFM(backbone=a.backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in T}, precision="bf16").to(dev)
eng = model.engine
eng._operands(dev)
cb = eng._opbatch
for _ in range(3):
    cb.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    cb.run()
e1.record()
torch.cuda.synchronize()
n = sum(p.numel() for p in model.parameters())
us = e0.elapsed_time(e1) / a.iters * 1e3
print(f"{a.backbone}: {n / 1e6:.1f} M parameters, {len(cb.jobs)} jobs, {us:.1f} us per rebuild ({12 * n / us / 1e6:.2f} TB/s at 12 B per parameter)")
