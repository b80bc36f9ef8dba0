"""Bit-reproducibility of the bench-size training step (DeiT-base + 5 heads, B = 128, bf16, weight-gradient side stream on): the same
step run 11 times must give bit-identical gradients (fixed-point LayerNorm sums, slab reductions in split order, no float atomics).

    python tools/determinism_check.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import theia_oracle as O
from test_model_gpu import build
bb, teachers, B = "facebook/deit-base-patch16-224", O.TEACHER_SETS["cddsv"], 128
model, _ = build(bb, teachers, "bf16")
images = O.synth_images(B, 0)
targets = {t: v.to("cuda:0") for t, v in O.synth_targets(B, teachers, 1).items()}
def step():
    model.zero_grad(set_to_none=True)
    losses = model.get_loss(model(images), targets)
    (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in model.named_parameters()}
ref = step()
bad = []
for it in range(10):
    g = step()
    bad.append(sum(not torch.equal(ref[k], g[k]) for k in ref))
print("bench-size step (B=128, bf16, side stream on) repeated 10x: parameters whose gradient differs from the first run:", bad)
