#!/usr/bin/env python
"""Micro-benchmark of the whole-sample LayerNorm passes at the bench shapes (b = 128, C = 768, 16x16 / 31x31 / 64x64 maps, bf16): HIP-event time
per launch and the algorithmic bytes per second.   python tools/chw_bench.py [--b 128] [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=128)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")


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


for hw in (16, 31, 64):
    C, b = 768, a.b
    E = hw * hw * C
    x = torch.randn(b, E, device=dev).bfloat16()
    dy = torch.randn(b, E, device=dev).bfloat16()
    g, be = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    x64 = x.double()
    sums = torch.stack([(x64.sum(1) * 2 ** 24).round(), ((x64 * x64).sum(1) * 2 ** 24).round()], 1).to(torch.int64).contiguous()
    del x64
    y, stats = ops.layernorm_chw_fwd(x, g, be, 1e-5, sums=sums)
    dg, db = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    t_apply = timeit(lambda: ops.layernorm_chw_fwd(x, g, be, 1e-5, sums=sums))
    t_bwd = timeit(lambda: ops.layernorm_chw_bwd(dy, x, g, stats, dg, db, relu_mask=True, accumulate=False))
    n = b * E
    print(f"{hw}x{hw}: apply (sums form) {t_apply:7.1f} us  {4 * n / t_apply / 1e6:5.2f} TB/s (4 B/elem)   backward (partial + finalize + bwd + reduces) {t_bwd:7.1f} us  "
          f"{10 * n / t_bwd / 1e6:5.2f} TB/s (10 B/elem)")
