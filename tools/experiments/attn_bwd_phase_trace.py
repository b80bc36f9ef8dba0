"""Per-workgroup phase timestamps of attn_bwd_fused_kernel (library built with -DAF_TRACE, see attn_bwd_phase_trace_not_kept.patch):
where the ~20 us per head go.  python tools/experiments/attn_bwd_phase_trace.py [--b 128 --h 12]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from theia_amd import ops, _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=128)
ap.add_argument("--h", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda:0")
n, D = 197, a.h * 64
qkv = torch.randn(a.b * n, 3 * D, device=dev).to(torch.bfloat16)
do = torch.randn(a.b * n, D, device=dev).to(torch.bfloat16)
o, lse = ops.attention_fwd(qkv, a.b, n, a.h)
for _ in range(3):
    ops.attention_bwd(qkv, o, do, lse, a.b, n, a.h)
torch.cuda.synchronize()
nb = a.b * a.h
buf = (ctypes.c_ulonglong * (nb * 8))()
lib = N.lib()
lib.theia_debug_attn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.theia_debug_attn_trace(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
us = lambda x: x / 100.0  # wall_clock64: 100 MHz
t0 = t[:, 0].min()
print(f"heads {nb}; kernel span {us(t[:, 4].max() - t0):.1f} us")
names = ["wait for the requested pieces + phase 0 (t1 - t0)", "phase 1 of wave 0 (t5 - t1)", "wait for the slowest wave (t2 - t5)", "K -> LDS (t3 - t2)", "phase 2 + dq stores issued (t4 - t3)"]
pairs = [(0, 1), (1, 5), (5, 2), (2, 3), (3, 4)]
for nm, (i, j) in zip(names, pairs):
    d = us(t[:, j] - t[:, i])
    print(f"  {nm:45s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
tot = us(t[:, 4] - t[:, 0])
print(f"  {'whole workgroup (t4 - t0)':45s} mean {tot.mean():6.2f}  p10 {np.percentile(tot, 10):6.2f}  p90 {np.percentile(tot, 90):6.2f} us")
# turnaround: group by (xcc, hw id without wave / simd bits) and look at the gap between one workgroup's end and the next one's start
key = (t[:, 7] & 0xf) * 65536 + ((t[:, 6] >> 8) & 0xff)
gaps, per_cu = [], []
for k in np.unique(key):
    rows = t[key == k]
    rows = rows[np.argsort(rows[:, 0])]
    per_cu.append(len(rows))
    gaps += list(us(rows[1:, 0] - rows[:-1, 4]))
gaps = np.array(gaps)
print(f"  distinct (xcc, se/sh/cu) ids {len(per_cu)}, workgroups per id {min(per_cu)}..{max(per_cu)}")
print(f"  gap between a head's last stamp and the next head's first on the same CU (the loop-end barrier): mean {gaps.mean():.2f}  p10 {np.percentile(gaps, 10):.2f}  p90 {np.percentile(gaps, 90):.2f} us")
