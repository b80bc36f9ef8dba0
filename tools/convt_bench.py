#!/usr/bin/env python
"""Stride-2 transposed convolutions of the translator heads (16->31 and 31->64, 768 channels, batch 128), one launch per output-parity
class: what the LayerNorm-statistics epilogue (ln_sums) costs next to the plain store, per class.
    python tools/convt_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd import ops, _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    dev, T, C, b = torch.device("cuda:0"), torch.bfloat16, 768, a.batch
    for name, IH, p, op in (("up31", 16, 1, 0), ("up64", 31, 0, 1)):
        plan = ops.plan_convT3x3(C, IH, 2, p, op)
        OH = plan.out_hw
        x = torch.randn(b, IH * IH * C, device=dev).to(T)
        w = (torch.randn(C, 9 * C, device=dev) * 0.02).to(T)
        bias = torch.randn(C, device=dev)
        out = torch.empty(b, OH * OH * C, dtype=T, device=dev)
        sums = torch.zeros(b, 2, dtype=torch.int64, device=dev)
        for ci, (rmap, mpi) in enumerate(plan.fwd):
            M, K = b * mpi, rmap.ntaps * C
            res = []
            for label, kw in (("plain", {}), ("relu", {"act": N.ACT_RELU}), ("sums", {"ln_sums": sums}), ("relu+sums", {"act": N.ACT_RELU, "ln_sums": sums})):
                def run():
                    ops.gemm_nt(x, w, out, M, C, K, rmap, 9 * C, C, bias=bias, **kw)
                kern = ops.KERNEL_NAMES.get(ops.gemm_nt(x, w, out, M, C, K, rmap, 9 * C, C, bias=bias, plan_only=True, **kw))
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / a.iters * 1e3
                res.append(f"{label} [{kern}] {us:7.1f} us {2.0 * M * C * K / us / 1e6:6.0f} TF")
            print(f"{name} class {ci}: M={M} N={C} K={K} taps={rmap.ntaps} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
