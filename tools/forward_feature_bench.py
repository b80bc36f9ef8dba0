#!/usr/bin/env python
"""forward_feature() inference throughput (BASELINE.json configs[4]: DeiT-base student, images streamed in chunks).

    python tools/forward_feature_bench.py [--backbone facebook/deit-base-patch16-224] [--chunk 512] [--chunks 8]
Synthetic uint8 images resident on the GPU, bf16, no_grad; prints images/s over --chunks chunks after 2 warm-up chunks."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd.models.rvfm import RobotVisionFM  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="facebook/deit-base-patch16-224")
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--chunks", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = RobotVisionFM(backbone=a.backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes=None, precision="bf16").to(dev).eval()
    imgs = torch.randint(0, 256, (a.chunk, 224, 224, 3), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        for _ in range(2):
            m.forward_feature(imgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.chunks):
            z = m.forward_feature(imgs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = a.chunk * a.chunks
    print(f"forward_feature {a.backbone.split('/')[-1]} bf16: {n / dt:.0f} images/s ({dt / a.chunks * 1e3:.1f} ms per chunk of {a.chunk}); out {tuple(z.shape)} {z.dtype}")


if __name__ == "__main__":
    main()
