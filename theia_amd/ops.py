"""Thin tensor-level wrappers over the C ABI (one Python function per kernel family) and the row-map builders
that express Linear / Conv3x3 / ConvTranspose3x3 (+ data/weight gradients) as implicit GEMMs.

Everything here launches HIP kernels on torch's current stream; nothing falls back to torch math.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as N
from ._native import RowMap, GemmArgs, WgradArgs

# ----------------------------------------------------------------------------------------------
# row maps
# ----------------------------------------------------------------------------------------------


def rowmap(taps: Sequence[Tuple[int, int, int]], rows_hw: Tuple[int, int], in_hw: Tuple[int, int], in_s: int, in_c: int,
           in_batch_stride: int, in_offset: int, out_w: int, out_s: int, out_y0: int, out_x0: int,
           out_batch_stride: int, out_offset: int) -> RowMap:
    """taps: (dy, dx, wslot) per tap."""
    m = RowMap()
    m.ntaps = len(taps)
    for t, (dy, dx, ws) in enumerate(taps):
        m.dy[t], m.dx[t], m.wslot[t] = dy, dx, ws
    m.rows_h, m.rows_w = rows_hw
    m.in_h, m.in_w = in_hw
    m.in_sy = m.in_sx = in_s
    m.in_c = in_c
    m.out_w, m.out_sy, m.out_sx, m.out_y0, m.out_x0 = out_w, out_s, out_s, out_y0, out_x0
    m.in_batch_stride, m.in_offset = in_batch_stride, in_offset
    m.out_batch_stride, m.out_offset = out_batch_stride, out_offset
    return m


def rm_plain(K: int, lda: int, ldo: int, a_offset: int = 0, o_offset: int = 0) -> RowMap:
    """Row-major [M,K] (pitch lda) -> [M,N] (pitch ldo)."""
    return rowmap([(0, 0, 0)], (1, 1), (1, 1), 1, K, lda, a_offset, 1, 1, 0, 0, ldo, o_offset)


TAPS9 = [(ky, kx) for ky in range(3) for kx in range(3)]


@dataclass
class ConvPlan:
    """Row maps of one 3x3 (transposed) convolution on NHWC activations.

    fwd: list of (rowmap, M_per_image) -- one entry, or 4 output-parity classes for stride-2 transposed convs
    dgrad: (rowmap, M_per_image)
    pack_fwd / pack_dgrad: (d0, d1, d2, s0, s1, s2) arguments of theia_cast_permute3 producing W[n][slot][c]
    grad_strides: (sn, ss, sc) of theia_wgrad_reduce writing into the reference weight layout
    wgrad_swapped: weight gradient as a reduction over INPUT pixels (conv_wgrad below) -- stride-2 transposed convs
    """
    fwd: List[Tuple[RowMap, int]]
    dgrad: Tuple[RowMap, int]
    pack_fwd: Tuple[int, int, int, int, int, int]
    pack_dgrad: Tuple[int, int, int, int, int, int]
    grad_strides: Tuple[int, int, int]
    out_hw: int
    wgrad_swapped: bool = False


def plan_conv3x3(C: int, H: int, in_bs: Optional[int] = None, in_off: int = 0) -> ConvPlan:
    """nn.Conv2d(C, C, 3, padding=1) on an HxH map; weight [co, ci, ky, kx] (adapter_heads.py:319-323)."""
    bs = H * H * C
    fwd = rowmap([(ky - 1, kx - 1, ky * 3 + kx) for ky, kx in TAPS9], (H, H), (H, H), 1, C, in_bs or bs, in_off, H, 1, 0, 0, bs, 0)
    dg = rowmap([(1 - ky, 1 - kx, ky * 3 + kx) for ky, kx in TAPS9], (H, H), (H, H), 1, C, bs, 0, H, 1, 0, 0, in_bs or bs, in_off)
    return ConvPlan([(fwd, H * H)], (dg, H * H), (C, 9, C, C * 9, 1, 9), (C, 9, C, 9, 1, C * 9), (C * 9, 1, 9), H)


def plan_convT3x3(C: int, IH: int, stride: int, padding: int, output_padding: int, in_bs: Optional[int] = None,
                  in_off: int = 0) -> ConvPlan:
    """nn.ConvTranspose2d(C, C, 3, stride, padding, output_padding); weight [ci, co, ky, kx]
    (adapter_heads.py:282-288 pad, :307-311 up-sampling).  out[i*s - p + ky] += in[i] * W[ky]."""
    OH = (IH - 1) * stride - 2 * padding + 3 + output_padding
    ibs = in_bs or IH * IH * C
    obs = OH * OH * C
    fwd: List[Tuple[RowMap, int]] = []
    if stride == 1:
        taps = [(padding - ky, padding - kx, ky * 3 + kx) for ky, kx in TAPS9]
        fwd.append((rowmap(taps, (OH, OH), (IH, IH), 1, C, ibs, in_off, OH, 1, 0, 0, obs, 0), OH * OH))
    else:
        assert stride == 2
        for py in range(2):
            for px in range(2):
                nyc, nxc = len(range(py, OH, 2)), len(range(px, OH, 2))
                taps = []
                for ky, kx in TAPS9:
                    if (py + padding - ky) % 2 == 0 and (px + padding - kx) % 2 == 0:
                        taps.append(((py + padding - ky) // 2, (px + padding - kx) // 2, ky * 3 + kx))
                fwd.append((rowmap(taps, (nyc, nxc), (IH, IH), 1, C, ibs, in_off, OH, 2, py, px, obs, 0), nyc * nxc))
    # d in[i,j] = sum_{ky,kx} d out[i*s - p + ky, j*s - p + kx] . W[ci, :, ky, kx]
    dg = rowmap([(ky - padding, kx - padding, ky * 3 + kx) for ky, kx in TAPS9], (IH, IH), (OH, OH), stride, C, obs, 0,
                IH, 1, 0, 0, ibs, in_off)
    return ConvPlan(fwd, (dg, IH * IH), (C, 9, C, 9, 1, C * 9), (C, 9, C, C * 9, 1, 9), (9, 1, C * 9), OH, stride == 2)


# ----------------------------------------------------------------------------------------------
# kernel wrappers
# ----------------------------------------------------------------------------------------------


def _dt(t: torch.Tensor) -> int:
    return N.dtype_code(t.dtype)


# Tile request applied to every theia_gemm_nt launch that does not pass its own (0 = the library chooses).  Test / self-check
# hook: the parity tests and bench.py's self-check run small batches through the persistent ping-pong kernel with 256256 (or
# 320256: its 320-row tiles), and cross-check the automatic choice at full size against the 2-stage kernel with 128128.  A
# ping-pong request is applied only to problems that kernel takes (theia_gemm_nt_plan decides); N < 64-wide problems keep the
# library's 128x64 choice.
GEMM_TILE_HINT = 0


def pp_supported(K: int, in_c: int, dtype: torch.dtype) -> bool:
    """requirements of the 256x256 ping-pong kernel (csrc/gemm.hip dispatch): half-tile granularity of K and in_c, and one
    tap's row within its 16 KiB zero page"""
    hkt = 32 if dtype == torch.bfloat16 else 16
    return K % hkt == 0 and in_c % hkt == 0 and in_c * (2 if dtype == torch.bfloat16 else 4) <= 16384


# when set to a list, gemm_nt appends (start_event, end_event, algorithmic_flops, tile_variant, (M, N, K)) per launch
GEMM_PROFILE: Optional[list] = None
WGRAD_PROFILE: Optional[list] = None  # same for theia_gemm_wgrad launches


def gemm_nt_algorithmic_bytes(M: int, Nn: int, K: int, rmap: RowMap, esz_in: int, esz_out: int, bias: bool, resid: bool, aux_in: bool,
                              aux_out: bool) -> int:
    """Minimum HBM bytes of one theia_gemm_nt launch: every operand element read once, every output element written once.
    Activations: a plain matrix has M x K elements; a multi-tap (convolution) row map gathers its K = taps x in_c columns from an input
    of ceil(M / rows per image) images x in_h x in_w x in_c elements (each pixel is read once however many taps touch it).  Weights:
    N x K.  Outputs: M x N (twice with a saved pre-activation).  Row inputs of the epilogue (residual, GELU' / ReLU' input): M x N each.
    Bias: 4 N."""
    R = rmap.rows_h * rmap.rows_w
    if rmap.ntaps > 1:
        act = min(M * K, -(-M // R) * rmap.in_h * rmap.in_w * rmap.in_c)
    else:
        act = M * rmap.in_c
    return (esz_in * (act + Nn * K) + esz_out * M * Nn * (1 + int(aux_out) + int(resid) + int(aux_in)) + (4 * Nn if bias else 0))


KERNEL_NAMES = {128128: "128x128", 128064: "128x64", 256000: "256x256-2stage", 256256: "256x256", 320256: "320x256", 256009: "256x256-conv"}


def gemm_nt(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, M: int, Nn: int, K: int, rmap: RowMap, ldw: int, ldo: int,
            bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, act: int = N.ACT_NONE,
            aux_in: Optional[torch.Tensor] = None, aux_out: Optional[torch.Tensor] = None,
            rowtab: Optional[torch.Tensor] = None, rowtab_period: int = 0, tile: int = 0, plan_only: bool = False,
            ln_sums: Optional[torch.Tensor] = None, scale_inv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, out8=None):
    """out8 = (e4m3 tensor like out, scale [1]): fp8 launches also write their output as e4m3 (theia_gemm_args_t.out8).
    tile: kernel request (0 = the library's choice; see theia_gemm_args_t.tile).  plan_only: launch nothing, return the code
    of the kernel the library would run (theia_gemm_nt_plan)."""
    g = GemmArgs()
    g.a, g.w, g.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias, g.resid, g.aux_in, g.aux_out = N.ptr(bias), N.ptr(resid), N.ptr(aux_in), N.ptr(aux_out)
    g.rowtab, g.rowtab_period = N.ptr(rowtab), rowtab_period
    g.M, g.N, g.K, g.ldw, g.ldo, g.act = M, Nn, K, ldw, ldo, act
    g.map = rmap
    g.ln_sums = N.ptr(ln_sums)  # int64 [images, 2] (2^-24 fixed point), zeroed by the caller: += (sum, sum of squares) per image
    assert ln_sums is None or ln_sums.dtype == torch.int64
    if a.dtype == torch.float8_e4m3fn:  # fp8 operands (quantize_fp8), bf16 output: de-quantisation factors as device scalars
        assert w.dtype == torch.float8_e4m3fn and out.dtype == torch.bfloat16
        if scale_inv is not None:
            g.a_scale_inv, g.w_scale_inv = scale_inv[0].data_ptr(), scale_inv[1].data_ptr()
        if out8 is not None:
            assert out8[0].dtype == torch.float8_e4m3fn and out8[0].numel() == out.numel()
            g.out8, g.out8_scale = out8[0].data_ptr(), out8[1].data_ptr()
        tile = 0 if tile in (0, 256256) else tile
    if tile == 0 and GEMM_TILE_HINT != 0 and a.dtype != torch.float8_e4m3fn:
        if GEMM_TILE_HINT == 128128:
            tile = 128128 if N.lib().theia_gemm_nt_tile(M, Nn, _dt(a)) != 128064 else 0
        elif N.lib().theia_gemm_nt_tile(M, Nn, _dt(a)) != 128064:  # 256256 / 320256: every ping-pong kernel the problem admits
            for req in (256009, GEMM_TILE_HINT):
                g.tile = req
                if N.lib().theia_gemm_nt_plan(g, _dt(a)) > 0:  # (a request the library cannot honour is an error code < 0)
                    tile = req
                    break
    g.tile = tile
    assert a.dtype == w.dtype and (a.dtype == out.dtype or a.dtype == torch.float8_e4m3fn)
    if plan_only:
        return N.lib().theia_gemm_nt_plan(g, _dt(a))
    if GEMM_PROFILE is not None:  # bench.py: HIP events on the launch stream around every theia_gemm_nt launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N.check(N.lib().theia_gemm_nt(g, _dt(a), N.stream_ptr()), "theia_gemm_nt")
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * M * Nn * K, KERNEL_NAMES.get(N.lib().theia_gemm_nt_plan(g, _dt(a)), "?"), (M, Nn, K),
                             gemm_nt_algorithmic_bytes(M, Nn, K, rmap, a.element_size(), out.element_size(), bias is not None,
                                                       resid is not None, aux_in is not None, aux_out is not None)))
        return out
    N.check(N.lib().theia_gemm_nt(g, _dt(a), N.stream_ptr()), "theia_gemm_nt")
    return out


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
           act: int = N.ACT_NONE, aux_in: Optional[torch.Tensor] = None, aux_out: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, tile: int = 0, scale_inv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
           out8=None) -> torch.Tensor:
    """out = act(x @ w.T + bias) + resid ; x [M,K], w [N,K] contiguous.  fp8 operands (float8_e4m3fn, with scale_inv) -> bf16 out."""
    M, K = x.shape
    Nn = w.shape[0]
    if out is None:
        out = torch.empty(M, Nn, dtype=torch.bfloat16 if x.dtype == torch.float8_e4m3fn else x.dtype, device=x.device)
    return gemm_nt(x, w, out, M, Nn, K, rm_plain(K, x.stride(0), out.stride(0)), w.stride(0), out.stride(0), bias, resid, act,
                   aux_in, aux_out, tile=tile, scale_inv=scale_inv, out8=out8)


def quantize_fp8(x: torch.Tensor, scale: torch.Tensor, amax: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [rows, C] (bf16 / f32, row stride x.stride(0)) -> float8_e4m3fn [rows, C] = e4m3(clamp(x * *scale, +-448)); *amax = max(*amax, max|x|).
    scale / amax: 1-element f32 device tensors (views into the engine's scale tables)."""
    rows, Cc = x.shape
    if out is None:
        out = torch.empty(rows, Cc, dtype=torch.float8_e4m3fn, device=x.device)
    N.check(N.lib().theia_quantize_fp8(x.data_ptr(), _dt(x), rows, Cc, x.stride(0), out.data_ptr(), scale.data_ptr(), N.ptr(amax), N.stream_ptr()),
            "theia_quantize_fp8")
    return out


class QuantBatch:
    """A table of fp8 quantisations of contiguous bf16 tensors executed by ONE launch (theia_quantize_fp8_batch): the e4m3 copies of
    every GEMM weight operand, rebuilt after each optimizer step.  add() while building, then run() any number of times; bound to the
    data pointers it was built with."""

    def __init__(self, device):
        self.device = device
        self.jobs: List[N.QuantJob] = []
        self._dev: Optional[torch.Tensor] = None
        self._blocks = 0
        self._keep: List[torch.Tensor] = []

    def add(self, src: torch.Tensor, dst: torch.Tensor, scale: torch.Tensor, amax: torch.Tensor) -> None:
        assert src.dtype == torch.bfloat16 and src.is_contiguous() and dst.is_contiguous() and dst.numel() == src.numel() and src.numel() % 8 == 0
        assert self._dev is None
        j = N.QuantJob()
        j.src, j.dst, j.scale, j.amax, j.n = src.data_ptr(), dst.data_ptr(), scale.data_ptr(), amax.data_ptr(), src.numel()
        self.jobs.append(j)
        self._keep += [src, dst, scale, amax]

    def run(self) -> None:
        if not self.jobs:
            return
        if self._dev is None:
            import ctypes
            arr = (N.QuantJob * len(self.jobs))(*self.jobs)
            self._blocks = N.lib().theia_quantize_fp8_batch_plan(ctypes.addressof(arr), len(self.jobs))
            if self._blocks <= 0:
                raise N.TheiaNativeError("theia_quantize_fp8_batch_plan: bad job table")
            raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8)
            self._dev = raw.to(self.device)
        N.check(N.lib().theia_quantize_fp8_batch(self._dev.data_ptr(), len(self.jobs), self._blocks, N.stream_ptr()), "theia_quantize_fp8_batch")


def fp8_update_scales(amax: torch.Tensor, scale: torch.Tensor, inv_scale: torch.Tensor, margin: float = 1.0) -> None:
    """delayed scaling: scale = 448 / (amax * margin) for every slot that saw data, inv_scale = 1 / scale, amax = 0"""
    N.check(N.lib().theia_fp8_update_scales(amax.data_ptr(), scale.data_ptr(), inv_scale.data_ptr(), amax.numel(), margin, N.stream_ptr()),
            "theia_fp8_update_scales")


def _wgrad_args(dy: torch.Tensor, a: torch.Tensor, slabs: torch.Tensor, M: int, Nn: int, ldo: int, kslots: int, splits: int,
                rmap: RowMap) -> WgradArgs:
    g = WgradArgs()
    g.dy, g.a, g.slabs = dy.data_ptr(), a.data_ptr(), slabs.data_ptr()
    g.M, g.N, g.ldo, g.kslots, g.splits = M, Nn, ldo, kslots, splits
    g.map = rmap
    return g


def gemm_wgrad(dy: torch.Tensor, a: torch.Tensor, slabs: torch.Tensor, M: int, Nn: int, ldo: int, kslots: int, splits: int,
               rmap: RowMap, bias_out: Optional[torch.Tensor] = None, bias_accumulate: bool = False,
               bias_slabs: Optional[torch.Tensor] = None, defer_bias: bool = False) -> bool:
    """slab[s][n][slot*in_c + c] = sum_{m in split s} dy[m, n] * a[m, (tap, c)].  With bias_out (f32 [N]) the bias gradient
    bias_out (+)= colsum(dy) is produced by the same launch when the kernel supports it (returns True); otherwise the
    caller has to run colsum (returns False)."""
    g = _wgrad_args(dy, a, slabs, M, Nn, ldo, kslots, splits, rmap)
    fused = False
    if bias_out is not None and bias_slabs is not None and N.lib().theia_wgrad_fuses_bias(g, _dt(dy)):
        assert bias_slabs.numel() >= splits * Nn and bias_out.dtype == torch.float32
        g.bias_slabs, g.bias_out, g.bias_accumulate = bias_slabs.data_ptr(), bias_out.data_ptr(), int(bias_accumulate)
        g.defer_bias_reduce = int(defer_bias)  # the bias partials are then reduced by wgrad_finish, in the weights' launch
        fused = True
    if WGRAD_PROFILE is not None:  # tuning aid (bench.py THEIA_BENCH_GEMM_TABLE): HIP events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N.check(N.lib().theia_gemm_wgrad(g, _dt(dy), N.stream_ptr()), "theia_gemm_wgrad")
        e1.record()
        WGRAD_PROFILE.append((e0, e1, 2.0 * M * Nn * kslots * rmap.in_c, f"s{splits}", (M, Nn, kslots * rmap.in_c)))
        return fused
    N.check(N.lib().theia_gemm_wgrad(g, _dt(dy), N.stream_ptr()), "theia_gemm_wgrad")
    return fused


def wgrad_splits(M: int, Nn: int, Ktot: int) -> int:
    return N.lib().theia_wgrad_splits(M, Nn, Ktot)


_DEVICE_CUS = 0


def device_cus() -> int:
    """CUs of the device (the GEMM planners' budget when nothing is reserved)"""
    global _DEVICE_CUS
    if _DEVICE_CUS == 0:
        cur = get_compute_cus()
        set_compute_cus(0)
        _DEVICE_CUS = get_compute_cus()
        if cur != _DEVICE_CUS:
            set_compute_cus(cur)
    return _DEVICE_CUS


def set_compute_cus(n: int) -> None:
    """CU budget of the GEMM planners (persistent NT grid, weight-gradient splits); 0 = the whole device.  Host-side state read when
    a launch is enqueued -- see theia_hip.h and parallel.TheiaDataParallel (CUs left to RCCL while gradients are exchanged)."""
    N.check(N.lib().theia_set_compute_cus(int(n)), "theia_set_compute_cus")


def get_compute_cus() -> int:
    return N.lib().theia_get_compute_cus()


def set_gemm_schedule(dynamic: bool) -> bool:
    """Tile schedule of the persistent NT GEMM: False = static rounds (default, fastest alone), True = work-conserving per-XCD queues
    (for launches that share the chip: RCCL channels at N > 1).  Returns the previous setting.  See theia_hip.h."""
    return bool(N.lib().theia_set_gemm_schedule(1 if dynamic else 0))


def get_gemm_schedule() -> bool:
    return bool(N.lib().theia_get_gemm_schedule())


def wgrad_reduce(slabs: torch.Tensor, splits: int, Nn: int, kslots: int, C: int, out: torch.Tensor, sn: int, ss: int, sc: int,
                 accumulate: bool) -> None:
    N.check(N.lib().theia_wgrad_reduce(slabs.data_ptr(), splits, Nn, kslots, C, out.data_ptr(), sn, ss, sc, int(accumulate),
                                       N.stream_ptr()), "theia_wgrad_reduce")


def wgrad_finish(slabs: torch.Tensor, splits: int, Nn: int, kslots: int, C: int, out: torch.Tensor, sn: int, ss: int, sc: int,
                 accumulate: bool, bias: Optional[Tuple[torch.Tensor, torch.Tensor, bool]] = None) -> None:
    """out (reference layout) (+)= sum over the split slabs, both sides coalesced; bias = (bias_slabs, bias_out, accumulate) reduces
    the bias partials of the same weight-gradient GEMM in the same launch."""
    bs, bo, ba = (bias[0].data_ptr(), bias[1].data_ptr(), int(bias[2])) if bias is not None else (None, None, 0)
    N.check(N.lib().theia_wgrad_finish(slabs.data_ptr(), splits, Nn, kslots, C, out.data_ptr(), sn, ss, sc, int(accumulate), bs, bo, ba,
                                       N.stream_ptr()), "theia_wgrad_finish")


def conv_wgrad_splits(plan: ConvPlan, b: int, C: int) -> int:
    """split count conv_wgrad will use (slab workspace = splits * 9 * C * C floats)."""
    mpi = plan.dgrad[1] if plan.wgrad_swapped else sum(m for _r, m in plan.fwd)
    return N.lib().theia_wgrad_splits_taps(b * mpi, C, 9, C)


def conv_wgrad(plan: ConvPlan, dy: torch.Tensor, x: torch.Tensor, b: int, C: int, grad_w: torch.Tensor, accumulate: bool,
               ws: Optional[torch.Tensor] = None, bias: Optional[Tuple[torch.Tensor, bool]] = None) -> None:
    """grad_w (reference layout, f32) (+)= weight gradient of one 3x3 (transposed) convolution; bias = (grad_b, accumulate)
    also produces the bias gradient colsum(dy) (inside the GEMM launch when it can, else with theia_colsum).
    dy: output gradient, flat NHWC [b, OH*OH*C]; x: the convolution's input (flat NHWC, possibly strided per the plan).

    Stride-2 transposed convolutions reduce over INPUT pixels: dW[ci, co, tap] = sum_{b,i,j} x[b,i,j,ci] * dy[b, 2i-p+ky, 2j-p+kx, co]
    -- theia_gemm_wgrad with the dgrad row map and the operands swapped (x is the dense operand, dy the gathered one).  Every
    tap then has the same number of rows and one launch fills the chip; reducing over OUTPUT pixels needs one launch per
    output-parity class with 1, 2, 2 and 4 live taps (27..108 workgroups each on 256 CUs)."""
    splits = conv_wgrad_splits(plan, b, C)
    need = splits * C * 9 * C
    extra = splits * C
    if ws is None or ws.numel() < need + extra:
        ws = torch.empty(need + extra, dtype=torch.float32, device=dy.device)
    slabs = ws[:need]
    fused = False
    if plan.wgrad_swapped:
        rmap, mpi = plan.dgrad
        gemm_wgrad(x, dy, slabs, b * mpi, C, C, 9, splits, rmap)
        wgrad_finish(slabs, splits, C, 9, C, grad_w, 9 * C, 1, 9, accumulate)  # slab[ci][tap][co] -> W[ci, co, ky, kx]
    else:
        for k, (rmap, mpi) in enumerate(plan.fwd):
            one = len(plan.fwd) == 1 and bias is not None
            fused = gemm_wgrad(dy, x, slabs, b * mpi, C, C, 9, splits, rmap, bias[0] if one else None, bias[1] if one else False,
                               ws[need:need + extra] if one else None, defer_bias=True)
        sn, ss, sc = plan.grad_strides
        wgrad_finish(slabs, splits, C, 9, C, grad_w, sn, ss, sc, accumulate, (ws[need:need + extra], bias[0], bias[1]) if fused else None)
    if bias is not None and not fused:
        colsum(dy.view(-1, C), bias[0], bias[1], ws)


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, grad_w: torch.Tensor, accumulate: bool,
                 ws: Optional[torch.Tensor] = None, bias: Optional[Tuple[torch.Tensor, bool]] = None) -> None:
    """grad_w[N,K] (+)= dy[M,N]^T @ x[M,K];  bias = (grad_b, accumulate): grad_b[N] (+)= colsum(dy), inside the GEMM launch
    when the kernel supports it, else with theia_colsum."""
    M, Nn = dy.shape
    K = x.shape[1]
    splits = wgrad_splits(M, Nn, K)
    need = splits * Nn * K
    extra = splits * Nn
    if ws is None or ws.numel() < need + extra:
        ws = torch.empty(need + extra, dtype=torch.float32, device=dy.device)
    fused = gemm_wgrad(dy, x, ws, M, Nn, dy.stride(0), 1, splits, rm_plain(K, x.stride(0), dy.stride(0)),
                       bias[0] if bias is not None else None, bias[1] if bias is not None else False, ws[need:need + extra], defer_bias=True)
    wgrad_finish(ws, splits, Nn, 1, K, grad_w, K, 0, 1, accumulate, (ws[need:need + extra], bias[0], bias[1]) if fused else None)
    if bias is not None and not fused:
        colsum(dy, bias[0], bias[1], ws)


_FINISH_GROUP = os.environ.get("THEIA_WGRAD_FINISH_GROUP", "1") != "0"  # A/B switch: 0 = one slab reduction per problem of a grouped launch


def linear_wgrad_group(problems, ws: Optional[torch.Tensor] = None) -> None:
    """The weight and bias gradients of several nn.Linear layers that share M, as ONE weight-gradient launch (theia_gemm_wgrad_group: the
    split count is CUs / (all their tiles), so a small problem no longer runs 28 splits of its own) + one slab reduction each.
    problems: [(dy [M, N_i], x [M, K_i], grad_w [N_i, K_i] f32, accumulate, (grad_b, accumulate))].  Falls back to linear_wgrad per
    problem when the library does not take the group (f32 parity mode, narrow shapes)."""
    M = problems[0][0].shape[0]
    ok = _dt(problems[0][0]) == N.BF16 and len(problems) <= 4 and all(dy.shape[0] == M and dy.dtype == problems[0][0].dtype
                                                                      for dy, *_ in problems)
    if ok:
        tiles = sum(N.lib().theia_wgrad_tiles(dy.shape[1], x.shape[1]) for dy, x, *_ in problems)
        splits = N.lib().theia_wgrad_group_splits(M, max(1, tiles))
        need = sum(splits * dy.shape[1] * (x.shape[1] + 1) for dy, x, *_ in problems)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.float32, device=problems[0][0].device)
        arr = (N.WgradArgs * len(problems))()
        parts = []
        off = 0
        for i, (dy, x, gw, accw, (gb, accb)) in enumerate(problems):
            Nn, K = dy.shape[1], x.shape[1]
            slabs = ws[off:off + splits * Nn * K]
            bslabs = ws[off + splits * Nn * K:off + splits * Nn * (K + 1)]
            off += splits * Nn * (K + 1)
            g = _wgrad_args(dy, x, slabs, M, Nn, dy.stride(0), 1, splits, rm_plain(K, x.stride(0), dy.stride(0)))
            if not N.lib().theia_wgrad_fuses_bias(g, _dt(dy)):
                ok = False
                break
            g.bias_slabs, g.bias_out, g.bias_accumulate, g.defer_bias_reduce = bslabs.data_ptr(), gb.data_ptr(), int(accb), 1
            arr[i] = g
            parts.append((slabs, bslabs, Nn, K, gw, accw, gb, accb))
    if ok:
        e0 = e1 = None
        if WGRAD_PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = N.lib().theia_gemm_wgrad_group(arr, len(problems), _dt(problems[0][0]), N.stream_ptr())
        if e0 is not None:
            e1.record()
        if rc == 0:
            if e0 is not None:
                WGRAD_PROFILE.append((e0, e1, sum(2.0 * M * q[2] * q[3] for q in parts), f"s{splits}g{len(parts)}",
                                      (M, sum(q[2] for q in parts), parts[0][3])))
            jobs = (N.WgradFinishJob * len(parts))()
            for i, (slabs, bslabs, Nn, K, gw, accw, gb, accb) in enumerate(parts):
                jobs[i] = N.WgradFinishJob(slabs.data_ptr(), gw.data_ptr(), bslabs.data_ptr(), gb.data_ptr(), K, Nn, K, int(accw), int(accb))
            rc = N.lib().theia_wgrad_finish_group(jobs, len(parts), splits, N.stream_ptr()) if len(parts) > 1 and _FINISH_GROUP else N.ERR_UNSUPPORTED
            if rc == N.ERR_UNSUPPORTED:  # (a gradient tensor that is not 16-byte aligned: one reduction per problem)
                for slabs, bslabs, Nn, K, gw, accw, gb, accb in parts:
                    wgrad_finish(slabs, splits, Nn, 1, K, gw, K, 0, 1, accw, (bslabs, gb, accb))
            else:
                N.check(rc, "theia_wgrad_finish_group")
            return
        if rc != N.ERR_UNSUPPORTED:
            N.check(rc, "theia_gemm_wgrad_group")
    for dy, x, gw, accw, bias in problems:
        linear_wgrad(dy, x, gw, accw, ws, bias=bias)


def colsum(x: torch.Tensor, out: torch.Tensor, accumulate: bool, ws: Optional[torch.Tensor] = None) -> None:
    M, Nn = x.shape
    need = N.lib().theia_colsum_workspace_bytes(M, Nn) // 4
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
    N.check(N.lib().theia_colsum(x.data_ptr(), M, Nn, x.stride(0), out.data_ptr(), ws.data_ptr(), int(accumulate), _dt(x),
                                 N.stream_ptr()), "theia_colsum")


def cast(src: torch.Tensor, dst: torch.Tensor) -> None:
    N.check(N.lib().theia_cast(src.data_ptr(), dst.data_ptr(), src.numel(), _dt(dst), N.stream_ptr()), "theia_cast")


def upcast_scale(src_bf16: torch.Tensor, dst_f32: torch.Tensor, scale: float = 1.0) -> None:
    """dst = float(src) * scale (flat buffers)"""
    assert src_bf16.dtype == torch.bfloat16 and dst_f32.dtype == torch.float32 and src_bf16.numel() == dst_f32.numel()
    N.check(N.lib().theia_upcast_scale_bf16(src_bf16.data_ptr(), dst_f32.data_ptr(), dst_f32.numel(), float(scale), N.stream_ptr()),
            "theia_upcast_scale_bf16")


def cast_transpose(src: torch.Tensor, dst: torch.Tensor, ldd: Optional[int] = None) -> None:
    R, Cc = src.shape
    N.check(N.lib().theia_cast_transpose(src.data_ptr(), dst.data_ptr(), R, Cc, ldd or R, _dt(dst), N.stream_ptr()),
            "theia_cast_transpose")


def cast_permute3(src: torch.Tensor, dst: torch.Tensor, d0: int, d1: int, d2: int, s0: int, s1: int, s2: int) -> None:
    N.check(N.lib().theia_cast_permute3(src.data_ptr(), dst.data_ptr(), d0, d1, d2, s0, s1, s2, _dt(dst), N.stream_ptr()),
            "theia_cast_permute3")


class CastBatch:
    """A table of permuting casts executed by ONE launch (theia_cast_batch): the per-step rebuild of every GEMM operand
    from the fp32 master weights.  add_*() while building, then run() any number of times; the table is bound to the data
    pointers it was built with."""

    def __init__(self, device, dtype: torch.dtype):
        self.device, self.dtype = device, dtype
        self.jobs: List[N.CastJob] = []
        self._dev: Optional[torch.Tensor] = None
        self._blocks = 0
        self._keep: List[torch.Tensor] = []

    def add(self, src: torch.Tensor, dst: torch.Tensor, d0: int, d1: int, d2: int, s0: int, s1: int, s2: int,
            t0: Optional[int] = None, t1: Optional[int] = None) -> None:
        """dst[i*t0 + j*t1 + k] = cast(src[i*s0 + j*s1 + k*s2]); default t: contiguous [d0, d1, d2]."""
        assert src.dtype == torch.float32 and dst.dtype in (self.dtype, torch.float32) and self._dev is None
        j = N.CastJob()
        j.src, j.dst = src.data_ptr(), dst.data_ptr()
        j.d0, j.d1, j.d2, j.dst_f32 = d0, d1, d2, int(dst.dtype == torch.float32 and self.dtype != torch.float32)
        j.s0, j.s1, j.s2 = s0, s1, s2
        j.t0, j.t1 = (d1 * d2 if t0 is None else t0), (d2 if t1 is None else t1)
        self.jobs.append(j)
        self._keep += [src, dst]

    def add_cast(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        """element-wise cast of a contiguous tensor (2-D view [rows, cols] when it has one)."""
        cols = src.shape[-1] if src.dim() >= 2 else src.numel()
        rows = src.numel() // cols
        self.add(src, dst, 1, rows, cols, 0, cols, 1)

    def add_transpose(self, src: torch.Tensor, dst: torch.Tensor, ldd: Optional[int] = None) -> None:
        """dst[c*ldd + r] = src[r, c] for a contiguous [R, C] matrix."""
        R, Cc = src.shape
        self.add(src, dst, 1, Cc, R, 0, 1, Cc, 0, ldd or R)

    def run(self) -> None:
        if not self.jobs:
            return
        if self._dev is None:
            import ctypes
            arr = (N.CastJob * len(self.jobs))(*self.jobs)
            self._blocks = N.lib().theia_cast_batch_plan(ctypes.addressof(arr), len(self.jobs))
            if self._blocks <= 0:
                raise N.TheiaNativeError("theia_cast_batch_plan: bad job table")
            raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8)
            self._dev = raw.to(self.device)
        N.check(N.lib().theia_cast_batch(self._dev.data_ptr(), len(self.jobs), self._blocks, N.dtype_code(self.dtype), N.stream_ptr()),
                "theia_cast_batch")


def unpermute3(src: torch.Tensor, dst: torch.Tensor, d0: int, d1: int, d2: int, t0: int, t1: int, t2: int,
               accumulate: bool) -> None:
    N.check(N.lib().theia_unpermute3_f32(src.data_ptr(), dst.data_ptr(), d0, d1, d2, t0, t1, t2, int(accumulate),
                                         N.stream_ptr()), "theia_unpermute3_f32")


def transpose_acc2(src0: torch.Tensor, dst0: torch.Tensor, acc0: bool, src1: torch.Tensor, dst1: torch.Tensor, acc1: bool, R: int, Cc: int) -> None:
    """transpose_acc of two [R, Cc] matrices by one launch"""
    N.check(N.lib().theia_transpose_acc2_f32(src0.data_ptr(), dst0.data_ptr(), int(acc0), src1.data_ptr(), dst1.data_ptr(), int(acc1), R, Cc,
                                             N.stream_ptr()), "theia_transpose_acc2_f32")


def transpose_acc(src: torch.Tensor, dst: torch.Tensor, R: int, Cc: int, accumulate: bool) -> None:
    """dst[c*R + r] (+)= src[r*C + c] (f32)"""
    N.check(N.lib().theia_transpose_acc_f32(src.data_ptr(), dst.data_ptr(), R, Cc, int(accumulate), N.stream_ptr()), "theia_transpose_acc_f32")


def resize_u8(img: torch.Tensor, channels_last: bool, out_h: int, out_w: int, resample: int = 2) -> torch.Tensor:
    """uint8 [b,H,W,3] / [b,3,H,W] on the GPU -> uint8 [b,out_h,out_w,3]: Image.resize((out_w, out_h), resample) of Pillow."""
    from .preprocess import resize_plan
    assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.dim() == 4
    b = img.shape[0]
    in_h, in_w = (img.shape[1], img.shape[2]) if channels_last else (img.shape[2], img.shape[3])
    pl = resize_plan(in_h, in_w, out_h, out_w, resample, img.device)
    out = torch.empty(b, out_h, out_w, 3, dtype=torch.uint8, device=img.device)
    tmp = torch.empty(b * pl.tmp_rows * out_w * 3, dtype=torch.uint8, device=img.device) if in_w != out_w and in_h != out_h else None
    N.check(N.lib().theia_resize_u8(img.data_ptr(), out.data_ptr(), N.ptr(tmp), b, in_h, in_w, int(channels_last), out_h, out_w,
                                    pl.bx.data_ptr(), pl.kx.data_ptr(), pl.ksize_x, pl.by.data_ptr(), pl.ky.data_ptr(), pl.ksize_y,
                                    pl.first_row, pl.tmp_rows, N.stream_ptr()), "theia_resize_u8")
    return out


def patchify(img: torch.Tensor, lut: torch.Tensor, out: torch.Tensor, channels_last: bool) -> None:
    """uint8 [b,H,W,3] / [b,3,H,W] -> [b*(H/16)*(W/16), 768] patch matrix through the preprocessing LUT"""
    b = img.shape[0]
    H, W = (img.shape[1], img.shape[2]) if channels_last else (img.shape[2], img.shape[3])
    N.check(N.lib().theia_patchify_u8_hw(img.data_ptr(), lut.data_ptr(), out.data_ptr(), b, H, W, int(channels_last), _dt(out),
                                         N.stream_ptr()), "theia_patchify_u8")


def write_cls(cls: torch.Tensor, pos: torch.Tensor, h: torch.Tensor, b: int, ntok: int, D: int) -> None:
    N.check(N.lib().theia_write_cls(cls.data_ptr(), pos.data_ptr(), h.data_ptr(), b, ntok, D, _dt(h), N.stream_ptr()),
            "theia_write_cls")


def write_tokens(tok: torch.Tensor, pos: torch.Tensor, h: torch.Tensor, b: int, ntok: int, t0: int, cnt: int, D: int) -> None:
    """h[b, t0 + r, :] = tok[r] + pos[r] (f32 [cnt, D] each)"""
    N.check(N.lib().theia_write_tokens(tok.data_ptr(), pos.data_ptr(), h.data_ptr(), b, ntok, t0, cnt, D, _dt(h), N.stream_ptr()),
            "theia_write_tokens")


def _q8(q8) -> Optional["N.Q8Out"]:
    """q8 = (e4m3 tensor shaped like the pass's bf16 output, scale [1] f32, amax [1] f32 or None) -> theia_q8_out_t (or None)"""
    if q8 is None:
        return None
    o = N.Q8Out()
    o.out, o.scale, o.amax = q8[0].data_ptr(), q8[1].data_ptr(), N.ptr(q8[2])
    assert q8[0].dtype == torch.float8_e4m3fn and q8[0].is_contiguous()
    return o


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, q8=None):
    """q8 (fp8 mode): the pass also writes y as e4m3 (theia_layernorm_fwd_q8) -- see _q8"""
    M, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    N.check(N.lib().theia_layernorm_fwd_q8(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), M, D, eps, _dt(x), _q8(q8), N.stream_ptr()), "theia_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dresid, dgamma, dbeta, accumulate: bool, ws: Optional[torch.Tensor] = None, q8=None):
    M, D = x.shape
    dx = torch.empty_like(x)
    need = N.lib().theia_layernorm_bwd_workspace_bytes(M, D) // 4
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
    N.check(N.lib().theia_layernorm_bwd_q8(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           N.ptr(dresid), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), M, D,
                                           int(accumulate), _dt(x), _q8(q8), N.stream_ptr()), "theia_layernorm_bwd")
    return dx


def layernorm_chw_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, ws: Optional[torch.Tensor] = None,
                      sums: Optional[torch.Tensor] = None, q8=None):
    """x [b, E] (NHWC flattened); gamma/beta f32 [E] in NHWC order.  sums: int64 [b, 2] fixed-point per-sample (sum, sum of squares) of x already
    accumulated by the producing GEMM's epilogue (gemm_nt(..., ln_sums=)) -> one pass instead of three."""
    b, E = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(b, 2, dtype=torch.float32, device=x.device)
    if sums is not None:
        N.check(N.lib().theia_layernorm_chw_fwd_sums_q8(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), sums.data_ptr(),
                                                        stats.data_ptr(), b, E, eps, _dt(x), _q8(q8), N.stream_ptr()), "theia_layernorm_chw_fwd_sums")
        return y, stats
    assert q8 is None, "the e4m3 output exists on the one-pass (sums) form only"
    need = N.lib().theia_layernorm_chw_workspace_bytes(b, E) // 4
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
    N.check(N.lib().theia_layernorm_chw_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                            ws.data_ptr(), b, E, eps, _dt(x), N.stream_ptr()), "theia_layernorm_chw_fwd")
    return y, stats


def layernorm_chw_bwd(dy, x, gamma, stats, dgamma, dbeta, relu_mask: bool, accumulate: bool, ws: Optional[torch.Tensor] = None,
                      dxsum: Optional[Tuple[torch.Tensor, bool]] = None, q8=None):
    """dxsum = (f32 [C], accumulate): also dxsum[c] (+)= sum over samples and pixels of dx -- the bias gradient of the convolution that
    produced x, from the pass that writes dx (theia_layernorm_chw_bwd_colsum)."""
    b, E = x.shape
    dx = torch.empty_like(x)
    need = N.lib().theia_layernorm_chw_workspace_bytes(b, E) // 4
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
    ds, dC, dacc = (dxsum[0].data_ptr(), dxsum[0].numel(), int(dxsum[1])) if dxsum is not None else (None, 0, 0)
    assert dxsum is None or dxsum[0].dtype == torch.float32
    N.check(N.lib().theia_layernorm_chw_bwd_colsum_q8(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), dx.data_ptr(),
                                                      dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), b, E, int(relu_mask),
                                                      int(accumulate), ds, dC, dacc, _dt(x), _q8(q8), N.stream_ptr()), "theia_layernorm_chw_bwd")
    return dx


def attention_fwd(qkv: torch.Tensor, b: int, n: int, h: int):
    D = h * 64
    o = torch.empty(b * n, D, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b * h * n, dtype=torch.float32, device=qkv.device)
    N.check(N.lib().theia_attention_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), b, n, h, _dt(qkv), N.stream_ptr()),
            "theia_attention_fwd")
    return o, lse


def attention_bwd(qkv, o, d_o, lse, b: int, n: int, h: int, ws: Optional[torch.Tensor] = None):
    dqkv = torch.empty_like(qkv)
    need = b * n * h
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=qkv.device)
    N.check(N.lib().theia_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                        ws.data_ptr(), b, n, h, _dt(qkv), N.stream_ptr()), "theia_attention_bwd")
    return dqkv


def distill_loss_fwd(pred: torch.Tensor, target: torch.Tensor, ws: Optional[torch.Tensor] = None):
    """pred [b, E] (compute dtype), target [b, E] f32 (or bf16 beside bf16 predictions) -> (losses f32[3] = (mse, cos, l1), coef f32 [b,2])."""
    b, E = pred.shape
    losses = torch.empty(3, dtype=torch.float32, device=pred.device)
    coef = torch.empty(b, 2, dtype=torch.float32, device=pred.device)
    need = N.lib().theia_distill_loss_workspace_bytes(b, E) // 4
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=pred.device)
    N.check(N.lib().theia_distill_loss_fwd_t(pred.data_ptr(), target.data_ptr(), _dt(target), losses.data_ptr(), coef.data_ptr(), ws.data_ptr(),
                                             b, E, _dt(pred), N.stream_ptr()), "theia_distill_loss_fwd")
    return losses, coef


def distill_loss_bwd(pred: torch.Tensor, target: torch.Tensor, coef: torch.Tensor, w: torch.Tensor, q8=None) -> torch.Tensor:
    b, E = pred.shape
    dpred = torch.empty_like(pred)
    N.check(N.lib().theia_distill_loss_bwd_q8(pred.data_ptr(), target.data_ptr(), _dt(target), coef.data_ptr(), w.data_ptr(), dpred.data_ptr(), b,
                                              E, _dt(pred), _q8(q8), N.stream_ptr()), "theia_distill_loss_bwd")
    return dpred


def token_select(x: torch.Tensor, b: int, n: int, D: int, disc: int, mode: int) -> torch.Tensor:
    shape = (b, n - 1 - disc, D) if mode == 0 else (b, D)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    N.check(N.lib().theia_token_select(x.data_ptr(), out.data_ptr(), b, n, D, disc, mode, _dt(x), N.stream_ptr()),
            "theia_token_select")
    return out


def feature_ingest_bf16(x_chw: torch.Tensor, mean: Optional[torch.Tensor], std: Optional[torch.Tensor],
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x_chw [b, C, H, W] (or [b, C, HW]) bf16 as stored on disk -> [b, HW, C] f32, normalised with the reference's bf16
    arithmetic when mean/std (f32 [C]) are given."""
    assert x_chw.dtype == torch.bfloat16 and x_chw.is_cuda and x_chw.is_contiguous() and x_chw.dim() in (3, 4)
    b, Cc = x_chw.shape[0], x_chw.shape[1]
    HW = x_chw.numel() // (b * Cc)
    if out is None:
        out = torch.empty(b, HW, Cc, dtype=torch.float32, device=x_chw.device)
    N.check(N.lib().theia_feature_ingest_bf16(x_chw.data_ptr(), N.ptr(mean), N.ptr(std), out.data_ptr(), b, Cc, HW, N.stream_ptr()),
            "theia_feature_ingest_bf16")
    return out


def feature_norm_bf16(x_bf16: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    rows, Cc = x_bf16.shape
    out = torch.empty(rows, Cc, dtype=torch.float32, device=x_bf16.device)
    N.check(N.lib().theia_feature_norm_bf16(x_bf16.data_ptr(), mean.data_ptr(), std.data_ptr(), out.data_ptr(), rows, Cc,
                                            N.stream_ptr()), "theia_feature_norm_bf16")
    return out


def add_inplace(dst: torch.Tensor, src: torch.Tensor) -> None:
    N.check(N.lib().theia_add_inplace(dst.data_ptr(), src.data_ptr(), dst.numel(), _dt(dst), N.stream_ptr()), "theia_add_inplace")


def fill_zero(t: torch.Tensor) -> None:
    N.check(N.lib().theia_fill_zero(t.data_ptr(), t.numel() * t.element_size(), N.stream_ptr()), "theia_fill_zero")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step: int, grad_scale: float = 1.0) -> None:
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    N.check(N.lib().theia_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, wd,
                                     bc1, bc2, grad_scale, N.stream_ptr()), "theia_adamw_step")


def adamw_step_scaled(p, g, m, v, lr, beta1, beta2, eps, wd, step: int, grad_scale_dev: torch.Tensor) -> None:
    """adamw_step with the gradient scale read from a 1-element device tensor (the clip coefficient)"""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    N.check(N.lib().theia_adamw_step_scaled(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, wd,
                                            bc1, bc2, grad_scale_dev.data_ptr(), N.stream_ptr()), "theia_adamw_step_scaled")


def adamw_step_dev(p, g, m, v, beta1, beta2, eps, wd, hyper_dev: torch.Tensor, grad_scale_dev: Optional[torch.Tensor] = None) -> None:
    """adamw_step with (lr, 1 - beta1^t, 1 - beta2^t) read from the 3-element f32 device tensor `hyper_dev` (capturable)"""
    N.check(N.lib().theia_adamw_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), beta1, beta2, eps, wd,
                                         hyper_dev.data_ptr(), N.ptr(grad_scale_dev), N.stream_ptr()), "theia_adamw_step_dev")


def grad_sumsq(g: torch.Tensor, partials: torch.Tensor) -> None:
    """partials[:theia_grad_sumsq_blocks()] = partial sums of squares of the flat f32 range g"""
    N.check(N.lib().theia_grad_sumsq(g.data_ptr(), g.numel(), partials.data_ptr(), N.stream_ptr()), "theia_grad_sumsq")


def grad_clip_coef(partials: torch.Tensor, max_norm: float, out2: torch.Tensor) -> None:
    """out2 = (total norm, min(1, max_norm / (total + 1e-6)))"""
    N.check(N.lib().theia_grad_clip_coef(partials.data_ptr(), partials.numel(), float(max_norm), out2.data_ptr(), N.stream_ptr()),
            "theia_grad_clip_coef")


def probe_tr16(image: torch.Tensor, addr: torch.Tensor) -> torch.Tensor:
    out = torch.empty(256, dtype=torch.int16, device=image.device)
    N.check(N.lib().theia_probe_tr16(image.data_ptr(), addr.data_ptr(), out.data_ptr(), N.stream_ptr()), "theia_probe_tr16")
    return out
