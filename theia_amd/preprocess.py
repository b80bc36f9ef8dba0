"""Host side of the image resize: filter-weight tables for ``theia_resize_u8``.

The reference's processor resizes through Pillow (``Image.resize``); the only floating-point part of that algorithm is the
computation of the per-output-pixel filter weights (``precompute_coeffs`` + ``normalize_coeffs_8bpc``, Pillow 12.2.0
src/libImaging/Resample.c), done in double precision on the host exactly as Pillow does; everything per pixel is integer
arithmetic on the GPU.  Tables are cached per (input size, output size, filter) and kept on the device."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

BILINEAR, BICUBIC = 2, 3  # PIL.Image.Resampling
PRECISION_BITS = 32 - 8 - 2
_SUPPORT = {BILINEAR: 1.0, BICUBIC: 2.0}


def _filter(x: np.ndarray, resample: int) -> np.ndarray:
    x = np.abs(x)
    if resample == BILINEAR:
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def resample_tables(in_size: int, out_size: int, resample: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """-> (bounds int32 [out, 2] = (first source index, tap count), weights int32 [out, ksize], ksize)"""
    if resample not in _SUPPORT:
        raise ValueError(f"resample={resample}: only BILINEAR (2) and BICUBIC (3) are implemented")
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = _SUPPORT[resample] * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    live = x < xmax[:, None]
    w = np.where(live, _filter(((x + xmin[:, None]) - center[:, None] + 0.5) * (1.0 / filterscale), resample), 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]  # left-to-right running sum, like the C loop (np.sum adds pairwise)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    fixed = np.trunc(np.where(w < 0, -0.5, 0.5) + w * float(1 << PRECISION_BITS)).astype(np.int32)
    fixed = np.where(live, fixed, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, fixed, ksize


class ResizePlan:
    """Device-resident tables for one (in_h, in_w) -> (out_h, out_w) resize."""

    def __init__(self, in_h: int, in_w: int, out_h: int, out_w: int, resample: int, device):
        self.in_h, self.in_w, self.out_h, self.out_w = in_h, in_w, out_h, out_w
        bx, kx, self.ksize_x = resample_tables(in_w, out_w, resample)
        by, ky, self.ksize_y = resample_tables(in_h, out_h, resample)
        self.first_row = int(by[0, 0])
        self.tmp_rows = int(by[-1, 0] + by[-1, 1]) - self.first_row
        if in_w != out_w:  # the vertical pass then reads the horizontal pass's image, which starts at first_row
            by = by.copy()
            by[:, 0] -= self.first_row
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.bx, self.kx, self.by, self.ky = to(bx), to(kx), to(by), to(ky)


_PLANS: Dict[tuple, ResizePlan] = {}


def resize_plan(in_h: int, in_w: int, out_h: int, out_w: int, resample: int, device) -> ResizePlan:
    key = (in_h, in_w, out_h, out_w, resample, str(device))
    if key not in _PLANS:
        _PLANS[key] = ResizePlan(in_h, in_w, out_h, out_w, resample, device)
    return _PLANS[key]
