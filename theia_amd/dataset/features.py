"""Teacher-feature ingest: the pre-extracted features a training step is fed with.

Reference path (host, per sample): ``decode_sample`` loads the safetensors blob, rearranges ``embedding`` from the on-disk
``[C, H, W]`` to ``(h w) c`` (dataset/data_utils.py:137-172), ``normalize_feature`` applies ``(x - mean) / std`` in bf16 with
the statistics cast to bf16 (:342-355, :374-379), and the training loop widens to fp32 (scripts/train/train_rvfm.py:112-114).
Here the host only parses the container and stages the raw bf16 bytes in pinned memory; the rearrange, the two bf16
roundings and the widening are one HIP kernel over the whole batch (``theia_feature_ingest_bf16``), fed by an asynchronous
H2D copy on its own stream -- 4.06 MB per image for the five teachers, the next bottleneck once compute is fast.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .. import ops


def load_feature_stats(dataset_root: str, feature_models: Iterable[str]) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(means, stds) per teacher, f32 [C].  File names as the reference's (data_utils.py:358-381): the ``imagenet_var_*``
    file holds the STANDARD DEVIATION (feature_extraction/calc_feature_mean.py:90), and it is used as such."""
    means: Dict[str, torch.Tensor] = {}
    stds: Dict[str, torch.Tensor] = {}
    for model in feature_models:
        name = model.replace("/", "_")
        means[model] = torch.from_numpy(np.load(os.path.join(dataset_root, f"imagenet_mean_{name}.npy"))).float().contiguous()
        stds[model] = torch.from_numpy(np.load(os.path.join(dataset_root, f"imagenet_var_{name}.npy"))).float().contiguous()
    return means, stds


def decode_feature(data: bytes) -> Dict[str, torch.Tensor]:
    """safetensors blob of one sample -> {"embedding": bf16 [C, H, W] (as stored), optionally "cls_token"} on the host.
    (The writer is feature_extraction_core/models.py:55-97; no rearrange here -- the GPU kernel does it.)"""
    from safetensors.torch import load as sft_load
    sft = sft_load(data)
    out = {"embedding": sft["embedding"].contiguous()}
    if "cls_token" in sft:
        out["cls_token"] = sft["cls_token"]
    return out


class FeatureIngest:
    """Batch of on-disk embeddings -> normalised fp32 ``[b, H*W, C]`` targets on the GPU (what ``get_loss`` consumes)."""

    def __init__(self, device, means: Optional[Dict[str, torch.Tensor]] = None, stds: Optional[Dict[str, torch.Tensor]] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FeatureIngest runs on a ROCm GPU only (no CPU fallback)")
        self.means = {k: v.to(self.device, torch.float32).contiguous() for k, v in (means or {}).items()}
        self.stds = {k: v.to(self.device, torch.float32).contiguous() for k, v in (stds or {}).items()}
        # Per teacher a ring of pinned staging buffers, each with the event of the H2D copy that last read it: the host runs
        # steps ahead of the GPU (the training loop only synchronises every log_interval steps), so a buffer may be rewritten
        # only after ITS copy has completed -- event.synchronize() before the host write; with two buffers that wait is
        # normally already over.
        self._ring: Dict[str, List[list]] = {}
        self._next: Dict[str, int] = {}
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self.ring_depth = 2

    def _stage(self, key: str, samples: List[torch.Tensor]):
        shape = (len(samples),) + tuple(samples[0].shape)
        ring = self._ring.get(key)
        if ring is None or tuple(ring[0][0].shape) != shape:
            for slot in ring or []:
                if slot[1] is not None:
                    slot[1].synchronize()
            ring = [[torch.empty(shape, dtype=torch.bfloat16).pin_memory(), None] for _ in range(self.ring_depth)]
            self._ring[key], self._next[key] = ring, 0
        slot = ring[self._next[key]]
        self._next[key] = (self._next[key] + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()  # the copy that last read this buffer has finished
        buf = slot[0]
        for i, s in enumerate(samples):
            if s.dtype != torch.bfloat16 or tuple(s.shape) != shape[1:]:
                raise ValueError(f"{key}: every embedding must be bf16 {shape[1:]}, got {s.dtype} {tuple(s.shape)}")
            buf[i].copy_(s)
        return slot

    def __call__(self, batch: Dict[str, List[torch.Tensor]]) -> Dict[str, torch.Tensor]:
        """batch[teacher] = list of bf16 [C, H, W] host tensors (one per sample) or one bf16 [b, C, H, W] tensor."""
        out: Dict[str, torch.Tensor] = {}
        cur = torch.cuda.current_stream(self.device)
        for key, val in batch.items():
            if isinstance(val, torch.Tensor) and val.is_cuda:
                dev_x = val.contiguous()
            else:
                slot = self._stage(key, list(val) if not isinstance(val, torch.Tensor) else list(val.unbind(0)))
                # the copy does not wait for compute already queued on `cur` (the destination is a fresh allocation of the
                # copy stream): H2D overlaps the previous step's kernels
                with torch.cuda.stream(self._copy_stream):
                    dev_x = slot[0].to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                slot[1] = ev
                cur.wait_event(ev)
                dev_x.record_stream(cur)
            out[key] = ops.feature_ingest_bf16(dev_x, self.means.get(key), self.stds.get(key))
        return out
