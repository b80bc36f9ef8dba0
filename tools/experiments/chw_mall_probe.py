"""Does the 256 MB memory-side cache help the LayerNorm[C,H,W] passes when the batch is processed in chunks?  Per-image time of the
backward (statistics pass + dx pass) and of the one-pass forward, right after a kernel that WROTE its input (as the producing GEMM does),
for chunk sizes 4 .. 128 images of a 64 x 64 x 768 map (6.3 MB per image and tensor)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from theia_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
E, C = 4096 * 768, 768
g = torch.ones(E, device=dev)
beta = torch.zeros(E, device=dev)
for b in (4, 8, 16, 32, 64, 128):
    src = torch.randn(b, E, device=dev).bfloat16()
    x = torch.relu(torch.randn(b, E, device=dev)).bfloat16()
    dy = torch.empty_like(src)
    _y, stats = ops.layernorm_chw_fwd(x, g, beta, 1e-5)
    dg, db = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    cs = torch.zeros(C, device=dev)
    ws = torch.empty(ops.N.lib().theia_layernorm_chw_workspace_bytes(b, E) // 4, device=dev)
    tb = tf = 0.0
    it = 12
    for i in range(it + 2):
        dy.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.layernorm_chw_bwd(dy, x, g, stats, dg, db, True, False, ws, dxsum=(cs, False))
        e1.record()
        dy.copy_(src)  # "x" of a forward LN, just produced
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        ops.layernorm_chw_fwd(dy, g, beta, 1e-5, ws)
        f1.record()
        torch.cuda.synchronize()
        if i >= 2:
            tb += e0.elapsed_time(e1)
            tf += f0.elapsed_time(f1)
    mb = E * 2 / 1e6
    print(f"b={b:4d} ({b * mb:6.0f} MB per tensor): backward {tb / it / b * 1e3:7.1f} us/image  (5 tensor passes -> {5 * mb / (tb / it / b * 1e3) / 1e6 * 1e6 / 1e3:5.2f} TB/s)"
          f"   3-pass forward {tf / it / b * 1e3:7.1f} us/image", flush=True)
