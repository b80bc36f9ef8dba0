#!/usr/bin/env python
"""Micro-benchmark of the row LayerNorm kernels at the bench shapes (M = b * 197 rows, D = 768 / 384 / 192, bf16): HIP-event time per launch.

    python tools/ln_bench.py [--b 128] [--iters 50]"""
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
    for D, b in ((768, a.b), (384, 2 * a.b), (192, 2 * a.b)):
        M = b * 197
        x = torch.randn(M, D, device=dev).bfloat16()
        dy = torch.randn(M, D, device=dev).bfloat16()
        dres = torch.randn(M, D, device=dev).bfloat16()
        g = torch.ones(D, device=dev)
        be = torch.zeros(D, device=dev)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        _y, mean, rstd = ops.layernorm_fwd(x, g, be, 1e-12)
        ws = torch.empty(ops.N.lib().theia_layernorm_bwd_workspace_bytes(M, D) // 4, device=dev)

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

        mb = M * D * 2 / 1e6
        tf = timeit(lambda: ops.layernorm_fwd(x, g, be, 1e-12))
        tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dg, db, False, ws))
        tb0 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, None, dg, db, False, ws))
        print(f"row LN M={M} D={D}: fwd {tf:6.1f} us ({2 * mb / tf / 1e6 * 1e3:5.2f} TB/s)   bwd+resid {tb:6.1f} us ({4 * mb / tb / 1e6 * 1e3:5.2f} TB/s)"
              f"   bwd {tb0:6.1f} us ({3 * mb / tb0 / 1e6 * 1e3:5.2f} TB/s)   [incl. allocation of the output and, bwd, the partial reduction]", flush=True)


if __name__ == "__main__":
    main()
