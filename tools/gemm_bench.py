#!/usr/bin/env python
"""Micro-benchmark of theia_gemm_nt / theia_gemm_wgrad on the hot-path shapes (tuning + rocprofv3 --pmc target).

    python tools/gemm_bench.py [--what nt|wgrad|both] [--iters 20] [--shapes conv16,fc1,...] [--tile 0|128128|256256|256009]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd import ops  # noqa: E402

SHAPES = {  # name: (M, N, K, kind)
    "conv16": (32768, 768, 6912, "conv"),
    "conv16d": (32768, 768, 6912, "conv_dgrad"),
    "pad": (32768, 768, 6912, "pad"),
    "fc1": (25216, 3072, 768, "plain"),
    "fc2": (25216, 768, 3072, "plain"),
    "proj": (25216, 768, 768, "plain"),
    "qkv": (25216, 2304, 768, "plain"),
    "qkvd": (25216, 768, 2304, "plain"),
    "up64c": (131072, 768, 3072, "plain"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="both")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--tile", type=int, default=0, help="kernel request for theia_gemm_nt (0 = library's choice)")
    ap.add_argument("--fp8", action="store_true", help="fp8 e4m3 operands (plain shapes, nt only)")
    ap.add_argument("--mnk", default="", help='extra plain shapes "M,N,K;M,N,K;..." (replaces --shapes)')
    a = ap.parse_args()
    if a.mnk:
        SHAPES.clear()
        for i, t in enumerate(a.mnk.split(";")):
            M_, N_, K_ = (int(v) for v in t.split(","))
            SHAPES[f"s{i}"] = (M_, N_, K_, "plain")
        a.shapes = ",".join(SHAPES)
    dev = torch.device("cuda:0")
    T = torch.bfloat16
    for name in a.shapes.split(","):
        M, N, K, kind = SHAPES[name]
        C = 768
        if kind in ("conv", "conv_dgrad", "pad"):
            b = M // 256
            if kind == "pad":
                plan = ops.plan_convT3x3(C, 14, 1, 0, 0, in_bs=197 * C, in_off=C)
                x = torch.randn(b, 197 * C, device=dev).to(T)
            else:
                plan = ops.plan_conv3x3(C, 16)
                x = torch.randn(b, 256 * C, device=dev).to(T)
            rmap, mpi = plan.dgrad if kind == "conv_dgrad" else plan.fwd[0]
            w = (torch.randn(N, K, device=dev) * 0.02).to(T)
            out = torch.empty(M, N, dtype=T, device=dev)

            def run_nt():
                ops.gemm_nt(x, w, out, M, N, K, rmap, K, N, tile=a.tile)
            kern = ops.KERNEL_NAMES.get(ops.gemm_nt(x, w, out, M, N, K, rmap, K, N, tile=a.tile, plan_only=True))
            dy = torch.randn(M, N, device=dev).to(T)
            splits = ops.wgrad_splits(M, N, K)
            slabs = torch.empty(splits * N * K, dtype=torch.float32, device=dev)

            def run_wg():
                ops.gemm_wgrad(dy, x, slabs, M, N, N, 9, splits, rmap)
        else:
            x = torch.randn(M, K, device=dev).to(T)
            w = (torch.randn(N, K, device=dev) * 0.02).to(T)
            out = torch.empty(M, N, dtype=T, device=dev)

            if a.fp8:
                one = torch.ones(1, device=dev)
                x8, w8 = ops.quantize_fp8(x, one), ops.quantize_fp8(w, one * 20)

                def run_nt():
                    ops.linear(x8, w8, out=out, scale_inv=(one, one))
                kern = "256x256-fp8"
            else:
                def run_nt():
                    ops.linear(x, w, out=out, tile=a.tile)
                kern = ops.KERNEL_NAMES.get(a.tile or ops.N.lib().theia_gemm_nt_tile(M, N, 1))
            dy = torch.randn(M, N, device=dev).to(T)
            g = torch.empty(N, K, dtype=torch.float32, device=dev)
            ws = torch.empty(ops.wgrad_splits(M, N, K) * N * K, dtype=torch.float32, device=dev)

            def run_wg():
                ops.linear_wgrad(dy, x, g, False, ws)
        for label, fn in (("nt", run_nt), ("wgrad", run_wg)):
            if a.what not in (label, "both") or (label == "wgrad" and (kind in ("conv_dgrad", "pad") or a.fp8)):
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            print(f"{label:5s} {name:7s} M={M} N={N} K={K} [{kern if label == 'nt' else 'wgrad'}]: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
