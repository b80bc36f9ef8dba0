"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the image resize the reference's HF image processor performs.

The reference calls ``self.processor(x, do_resize=True, ...)`` (models/backbones.py:337-339); the processor
(transformers ``ViTImageProcessorPil``: size 224x224, ``resample=BILINEAR``; DeiT's class default is BICUBIC) converts each
image to PIL and calls ``Image.resize((w, h), resample)``.  That algorithm lives in a third-party dependency that is not
part of /root/reference: **Pillow 12.2.0**, ``src/libImaging/Resample.c`` (``ImagingResample``, 8 bits per channel).  It is
restated here from its published source:

  * ``precompute_coeffs``: per output pixel a window [xmin, xmin+n) of source pixels and double-precision filter weights
    (triangle / Keys a=-0.5 cubic), the filter stretched by the down-scaling factor (antialiasing), weights normalised;
  * ``normalize_coeffs_8bpc``: weights -> fixed point, PRECISION_BITS = 32 - 8 - 2 = 22, round half away from zero;
  * horizontal pass over the source rows the vertical pass needs, rounded to uint8 (accumulator starts at 1 << 21,
    ``>> 22``, clipped to [0, 255]); then the vertical pass on that intermediate image.

Pinned (tests/test_resize_cpu.py) against Pillow itself run in this container on many size pairs, and against golden
vectors produced by the reference's own processor class (tests/golden/g12_*.npz, oracle/gen_golden.py)."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = 2, 3  # PIL.Image.Resampling values


def _bilinear(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def precompute_coeffs(in_size: int, out_size: int, resample: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """-> (bounds int32 [out, 2] = (first source index, tap count), weights int32 [out, ksize] in 22-bit fixed point, ksize)"""
    filt, fsupport = _FILTERS[resample]
    in0, in1 = 0.0, float(in_size)
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)          # C (int) cast: truncation toward zero
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(src: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """one resampling pass along `axis` of a uint8 [H, W, C] image"""
    src = np.moveaxis(src, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        # INT32 accumulator in the original: values stay far inside the range (|sum of weights| ~ 2^22, pixels <= 255)
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img: np.ndarray, out_h: int, out_w: int, resample: int = BILINEAR) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [out_h, out_w, C]; Image.resize((out_w, out_h), resample) of Pillow 12.2.0."""
    assert img.dtype == np.uint8 and img.ndim == 3
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    bh, kh, _ = precompute_coeffs(in_w, out_w, resample)
    bv, kv, _ = precompute_coeffs(in_h, out_h, resample)
    cur = img
    if in_w != out_w:  # horizontal pass first, only over the source rows the vertical pass reads
        first = int(bv[0, 0])
        last = int(bv[-1, 0] + bv[-1, 1])
        cur = _pass(img[first:last], bh, kh, axis=1)
        bv = bv.copy()
        bv[:, 0] -= first
    if in_h != out_h:
        cur = _pass(cur, bv, kv, axis=0)
    return cur
