#!/usr/bin/env python
"""Micro-benchmark of theia_distill_loss_fwd / _bwd at the bench's teacher shapes (b = 128, 256 tokens x Ct, bf16 predictions, f32 targets).

    python tools/loss_bench.py [--b 128] [--iters 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=128)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    w = torch.tensor([0.0, 0.9, 0.1], device=dev)
    for E in (256 * 1280, 256 * 1024, 256 * 64 * 64, 256 * 64):  # ViT-H / large teachers (256 tokens x Ct), the SAM map (256 x 64 x 64), a small one
        pred = torch.randn(a.b, E, device=dev).bfloat16()
        tgt = torch.randn(a.b, E, device=dev)
        ws = torch.empty(ops.N.lib().theia_distill_loss_workspace_bytes(a.b, E) // 4, device=dev)
        _l, coef = ops.distill_loss_fwd(pred, tgt, ws)

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters * 1e3

        n = a.b * E
        tf = timeit(lambda: ops.distill_loss_fwd(pred, tgt, ws))
        tb = timeit(lambda: ops.distill_loss_bwd(pred, tgt, coef, w))
        print(f"loss b={a.b} E={E}: fwd (partial + finalize) {tf:6.1f} us = {6.0 * n / tf / 1e6:5.2f} TB/s   bwd {tb:6.1f} us = {8.0 * n / tb / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
