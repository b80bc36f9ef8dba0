#!/usr/bin/env python
"""Micro-benchmark of theia_attention_fwd / _bwd at the bench shape (b=128, n=197, h=12, bf16): HIP-event time per launch.

    python tools/attn_bench.py [--b 128] [--n 197] [--h 12] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=128)
    ap.add_argument("--n", type=int, default=197)
    ap.add_argument("--h", type=int, default=12)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D = a.h * 64
    qkv = torch.randn(a.b * a.n, 3 * D, device=dev).to(torch.bfloat16)
    do = torch.randn(a.b * a.n, D, device=dev).to(torch.bfloat16)
    o, lse = ops.attention_fwd(qkv, a.b, a.n, a.h)
    ws = torch.empty(a.b * a.n * a.h, dtype=torch.float32, device=dev)
    flops_fwd = 4.0 * a.n * a.n * 64 * a.b * a.h

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

    us = timeit(lambda: ops.attention_fwd(qkv, a.b, a.n, a.h))
    print(f"attention fwd  b={a.b} n={a.n} h={a.h}: {us:8.1f} us  {flops_fwd / us / 1e6:7.1f} TF", flush=True)
    us = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, a.b, a.n, a.h, ws))
    print(f"attention bwd  b={a.b} n={a.n} h={a.h}: {us:8.1f} us  {2.5 * flops_fwd / us / 1e6:7.1f} TF (2.5x fwd flops)", flush=True)


if __name__ == "__main__":
    main()
