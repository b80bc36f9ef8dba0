"""StudentEngine: drives the HIP kernels for the Theia hot path (forward AND hand-written backward).

    uint8 images --patchify/LUT--> patch GEMM (+bias +pos) --> 12 x {LN, QKV GEMM, attention, out-proj GEMM(+res),
    LN, FC1 GEMM(+GELU), FC2 GEMM(+res)} --> LN --> per-teacher heads {pad ConvT, LN_chw, conv/convT(+ReLU) x2,
    LN_chw, Linear} --> distillation loss

The engine owns no parameters: it reads the fp32 master parameters held by ``RobotVisionFM`` (reference state_dict
layout), derives the operand layouts the kernels want (bf16/f32 casts, [co][tap][ci] conv packs, transposes for
data-gradients) into an operand cache that is refreshed when a parameter changes, and writes gradients straight
into flat per-bucket gradient buffers (``param.grad`` become views of them) so that data-parallel all-reduce can
run bucket-by-bucket while the rest of backward is still executing.

Autograd sees three nodes (backbone, translator, per-teacher loss); inside a node everything is explicit HIP
launches on the current stream -- there is no torch math and no CPU fallback on the product path.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N
from . import ops
from .models.backbones import GRID, IMAGE, MAX_TOKENS, NUM_LAYERS

# bumped by optimizers that update parameters through raw pointers (torch's version counter does not see those)
PARAM_EPOCH = [0]

LN_EPS_VIT = 1e-12   # ViTConfig.layer_norm_eps
LN_EPS_HEAD = 1e-5   # nn.LayerNorm default (adapter_heads.py:306-324)


def preprocess_lut(do_rescale: bool, do_normalize: bool, mean: Sequence[float], std: Sequence[float]) -> np.ndarray:
    """[3,256] f32 table reproducing the HF processor arithmetic bit-for-bit (host side, 768 values):
    rescale = float32(float64(v) * (1/255)) (transformers image_transforms.py:118-122), normalize = float32
    (x - mean) / std with float32 mean/std (:419-439).  Reference call site: models/backbones.py:337-339."""
    v = np.arange(256, dtype=np.uint8)
    lut = np.zeros((3, 256), dtype=np.float32)
    for c in range(3):
        x = v
        if do_rescale:
            x = (x.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
        if do_normalize:
            if not np.issubdtype(x.dtype, np.floating):
                x = x.astype(np.float32)
            x = (x - np.array(mean[c], dtype=x.dtype)) / np.array(std[c], dtype=x.dtype)
        lut[c] = x.astype(np.float32)
    return lut


def to_uint8_batch(x: Any, any_size: bool = False):
    """Accepts what the reference's DeiT.forward accepts (uint8 torch [B,H,W,C] / [B,C,H,W], single image, numpy,
    list of PIL images / arrays).  Returns (uint8 tensor, channels_last) -- or, for a list whose items differ in size (only
    with any_size), a list of such pairs.  Channel layout inference follows transformers
    image_utils.infer_channel_dimension_format: a leading dim of 1/3 means channels-first.
    any_size=False: only 224x224 images (what the patch-embedding consumes); True: the caller resizes."""
    if isinstance(x, (list, tuple)):
        items = [to_uint8_batch(i, any_size) for i in x]
        cl = items[0][1]
        ts = [t if c == cl else (t.permute(0, 2, 3, 1) if cl else t.permute(0, 3, 1, 2)) for t, c in items]
        if len({tuple(t.shape[1:]) for t in ts}) > 1:
            return [(t.contiguous(), cl) for t in ts]
        return torch.cat(ts, 0).contiguous(), cl
    if not isinstance(x, (torch.Tensor, np.ndarray)):
        x = np.asarray(x)  # PIL
        if x.ndim == 2:
            x = np.stack([x] * 3, -1)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.dim() == 3:
        x = x.unsqueeze(0)
    if x.dim() != 4:
        raise ValueError(f"expected a 3-D or 4-D image tensor, got shape {tuple(x.shape)}")
    if x.dtype != torch.uint8:
        raise TypeError("the MI355X ingest kernel takes uint8 pixels in [0,255] (reference default input)")
    channels_last = not (x.shape[1] in (1, 3))
    if (channels_last and x.shape[-1] != 3) or (not channels_last and x.shape[1] != 3):
        raise ValueError(f"expected 3-channel images, got shape {tuple(x.shape)}")
    hh, ww = (x.shape[1], x.shape[2]) if channels_last else (x.shape[2], x.shape[3])
    if not any_size and (hh != IMAGE or ww != IMAGE):
        raise NotImplementedError(f"{hh}x{ww} input: feed {IMAGE}x{IMAGE} images")
    return x.contiguous(), channels_last


class Geometry:
    """Token layout of one forward pass: [CLS (tok0 = 1) | gh*gw patch tokens | nreg register tokens]."""

    def __init__(self, gh: int, gw: int, tok0: int, nreg: int, interp: bool):
        self.gh, self.gw, self.P, self.tok0, self.nreg, self.interp = gh, gw, gh * gw, tok0, nreg, interp
        self.ntok = tok0 + self.P + nreg
        self.key = (gh, gw, tok0, nreg, interp)


def bicubic_matrix(n_in: int, n_out: int, scale: float) -> np.ndarray:
    """[n_out, n_in] f32 matrix of torch's upsample_bicubic (align_corners=False, A = -0.75) along one axis, in its f32
    arithmetic: source coordinate scale*(dst + 0.5) - 0.5, 4 taps at floor-1 .. floor+2 with indices clamped to the border.
    (torch/aten UpSampleBicubic2d / UpSample.h cubic_convolution1/2.)"""
    A = np.float32(-0.75)
    m = np.zeros((n_out, n_in), dtype=np.float32)
    for o in range(n_out):
        real = np.float32(scale) * (np.float32(o) + np.float32(0.5)) - np.float32(0.5)
        i0 = int(np.floor(real))
        t = np.float32(real - np.float32(i0))

        def c1(x):
            return ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + np.float32(1)

        def c2(x):
            return ((A * x - np.float32(5) * A) * x + np.float32(8) * A) * x - np.float32(4) * A

        w = [c2(t + np.float32(1)), c1(t), c1(np.float32(1) - t), c2(np.float32(2) - t)]
        for k in range(4):
            m[o, min(max(i0 - 1 + k, 0), n_in - 1)] += w[k]
    return m


class Fp8Scales:
    """Per-tensor delayed scaling of the fp8 (e4m3) GEMM operands (precision="fp8", BASELINE configs[3]).  One slot per
    quantisation site (an activation / gradient tensor at a fixed place of the step, or a weight operand): the quantiser records
    max|x| into `amax` while it quantises with the scale derived from the PREVIOUS step's maximum; `update()` (once per step,
    when the operand cache is rebuilt after the optimizer) turns the maxima into the next scales on the device -- no host
    round trip.  A site's very first use calibrates itself with an extra max-only pass."""

    def __init__(self, device, capacity: int = 2048):
        self.amax = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.scale = torch.ones(capacity, dtype=torch.float32, device=device)
        self.inv = torch.ones(capacity, dtype=torch.float32, device=device)
        self.index: Dict[str, int] = {}
        self.calibrated: set = set()
        self.steps = 0                                                          # update() calls so far
        self.refresh = max(1, int(os.environ.get("THEIA_FP8_AMAX_EVERY", "32")))  # every refresh-th step records maxima (see fused())

    def slot(self, site: str) -> int:
        i = self.index.get(site)
        if i is None:
            i = self.index[site] = len(self.index)
            if i >= self.amax.numel():
                raise RuntimeError("Fp8Scales: out of slots")
        return i

    def quantize(self, x2d: torch.Tensor, site: str):
        """-> (float8_e4m3fn [rows, C], inv_scale view)"""
        i = self.slot(site)
        sc, am, inv = self.scale[i:i + 1], self.amax[i:i + 1], self.inv[i:i + 1]
        if site not in self.calibrated:  # first use: max-only pass, then this slot's scale
            q = ops.quantize_fp8(x2d, sc, am)
            ops.fp8_update_scales(am, sc, inv)
            self.calibrated.add(site)
            return ops.quantize_fp8(x2d, sc, am, out=q), inv
        # (between refresh steps the pass does not record the maximum either: its one atomic per block is ~10 us of a 12-50 us pass)
        return ops.quantize_fp8(x2d, sc, am if self.steps % self.refresh == 0 else None), inv

    def fused(self, site: str, like, records_max: bool = True):
        """(e4m3 buffer shaped like `like`, scale, amax, inv_scale) for a producer that writes its output's e4m3 copy itself (the *_q8
        passes: ops.layernorm_fwd(..., q8=) ...), or None while the site has not calibrated itself through quantize() yet (its first
        step) -- round 6: the separate 3-byte-per-element pass becomes one more byte written by the pass that produces the tensor"""
        # The fused passes do NOT record max|x|: a maximum per slot is one device-scope atomic per wave or block, performed at the memory side
        # and serialised per address (the effect that made the stand-alone pass take 90-105 us, above) -- with them the fused LayerNorm passes
        # were 0.4 ms per step SLOWER than the separate quantise passes.  Instead every `refresh`-th step runs the separate passes (which
        # record the maxima and so refresh the scales); in between the scales stay as they are -- delayed scaling with an update interval,
        # e4m3 saturation (+-448) under it as always.  THEIA_FP8_AMAX_EVERY (default 32).
        # On the refresh steps the producers that CAN record the maximum (the *_q8 passes) do, atomics and all -- a slower pass once in
        # `refresh` steps; the GEMM epilogues cannot (records_max=False): their consumers run the separate pass on those steps.
        refresh_step = self.steps % self.refresh == 0
        if site not in self.calibrated or (refresh_step and not records_max):
            return None
        i = self.slot(site)
        shape, dev = (like.shape, like.device) if isinstance(like, torch.Tensor) else like
        return [torch.empty(shape, dtype=torch.float8_e4m3fn, device=dev), self.scale[i:i + 1], self.amax[i:i + 1] if refresh_step else None,
                self.inv[i:i + 1]]

    def update(self, activations_only: bool = False) -> None:
        """amax -> next scale / inverse scale.  activations_only: leave the weight ("w:") slots alone -- their e4m3 copies in the
        operand cache were quantised with the scale in force when the cache was built and carry a VIEW of this `inv`; refreshing a
        weight slot without re-quantising the weight would rescale every output of that GEMM by amax_t / amax_{t-1}."""
        self.steps += 1
        n = len(self.index)
        if not n:
            return
        if not activations_only:
            ops.fp8_update_scales(self.amax[:n], self.scale[:n], self.inv[:n])
            return
        # slots are handed out in first-use order: maximal runs of consecutive activation slots, one launch each
        names = sorted(self.index, key=self.index.get)
        i = 0
        while i < n:
            if names[i].startswith("w:"):
                i += 1
                continue
            j = i
            while j < n and not names[j].startswith("w:"):
                j += 1
            ops.fp8_update_scales(self.amax[i:j], self.scale[i:j], self.inv[i:j])
            i = j


class _SideQueue:
    """A second HIP stream (+ its own scratch) for work whose results are only needed at the end of backward.

    ``run(fn, *tensors)`` enqueues ``fn`` behind everything issued so far on the current stream; the tensors it reads are
    pinned with ``record_stream`` so the caching allocator does not recycle them early; ``join()`` makes the current
    stream wait for the queue; ``fence()`` only returns an event (for a consumer on a third stream, e.g. the gradient
    all-reduce) and ``join_at_backward_end()`` defers the join to the end of the running autograd backward pass, so the
    main stream never idles on the queue in the middle of backward.  With ``enabled=False`` everything runs inline (A/B
    switch THEIA_SIDE_STREAM=0)."""

    def __init__(self, device, enabled: bool = True):
        self.device = device
        self.enabled = enabled
        # HIP exposes two stream priorities here (0 = default/low, -1 = high); the queue stays at the default.  A/B on the
        # bench (THEIA_SIDE_PRIORITY=-1 / 0): no difference, the CU share of co-running kernels is set by their LDS footprint.
        prio = int(os.environ.get("THEIA_SIDE_PRIORITY", "0"))
        self.stream = torch.cuda.Stream(device=device, priority=prio) if enabled else None
        self.ws: Optional[torch.Tensor] = None
        self._dirty = False
        self._cb_queued = False

    def ensure_ws(self, nfloats: int) -> None:
        if self.ws is None or self.ws.numel() < nfloats:
            self.join()
            self.ws = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=self.device)

    def run(self, fn, *tensors) -> None:
        if not self.enabled:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            fn()
        for t in tensors:
            t.record_stream(self.stream)
        self._dirty = True

    def join(self) -> None:
        if self.enabled and self._dirty:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._dirty = False

    def fence(self) -> Optional["torch.cuda.Event"]:
        """Event marking everything queued so far (None when there is nothing to wait for)."""
        if not (self.enabled and self._dirty):
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def join_at_backward_end(self) -> None:
        """join() once, when the autograd backward pass that is running finishes (queue_callback: the mechanism DDP uses
        for its final synchronisation); outside a backward pass, join now."""
        if not (self.enabled and self._dirty) or self._cb_queued:
            return
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self._backward_end)
            self._cb_queued = True
        except RuntimeError:  # "Final callbacks can only be installed during backward pass."
            self.join()

    def _backward_end(self) -> None:
        self._cb_queued = False
        self.join()


class GradBucket:
    """A flat fp32 gradient buffer for a group of parameters, filled in backward-completion order."""

    def __init__(self, name: str, named_params: List[Tuple[str, torch.nn.Parameter]]):
        from .optimizers.utils import is_no_decay
        self.name = name
        # [weight-decayed params | no-decay params] so a fused optimizer needs two launches per bucket
        decay = [(n, p) for n, p in named_params if not is_no_decay(n, p)]
        nodecay = [(n, p) for n, p in named_params if is_no_decay(n, p)]
        self.names = [n for n, _ in decay + nodecay]
        self.params = [p for _, p in decay + nodecay]
        self.offsets: List[int] = []
        off = 0
        self.decay_numel = 0
        for i, p in enumerate(self.params):
            if i == len(decay):
                self.decay_numel = off
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8  # keep every view 32-byte aligned for vector stores
        if len(nodecay) == 0:
            self.decay_numel = off
        self.numel = off
        self.flat: Optional[torch.Tensor] = None
        self._views: Dict[int, torch.Tensor] = {}   # parameter index -> its view of `flat` (built once per buffer: 240 lookups per step)
        self._views_of: Optional[torch.Tensor] = None

    def ensure(self, device) -> torch.Tensor:
        if self.flat is None or self.flat.device != device:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        return self.flat

    def view(self, i: int) -> torch.Tensor:
        if self._views_of is not self.flat:  # (the optimizer / a checkpoint load may have re-pointed the buffer)
            self._views, self._views_of = {}, self.flat
        v = self._views.get(i)
        if v is None:
            p = self.params[i]
            v = self._views[i] = self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)
        return v


class StudentEngine:
    def __init__(self, rvfm, precision: str = "fp32"):
        if precision not in ("fp32", "bf16", "fp8"):
            raise ValueError("precision must be 'fp32' (exact-f32 MFMA, parity mode), 'bf16' (throughput mode) or 'fp8' (bf16 "
                             "activations, e4m3 operands for the forward / data-gradient GEMMs, bf16 weight-gradient GEMMs)")
        self.rvfm = rvfm
        self.dtype = torch.float32 if precision == "fp32" else torch.bfloat16
        self.precision = precision
        self.fp8: Optional[Fp8Scales] = None  # created with the operand table (needs the device)
        vit = rvfm.backbone.model
        self.D, self.heads, self.F = vit.hidden_size, vit.num_heads, vit.intermediate_size
        self.tok0, self.nreg = (1 if vit.has_cls else 0), vit.num_reg_tokens  # token layout of the student (nocls- / reg-)
        self.geo224 = Geometry(GRID, GRID, self.tok0, self.nreg, False)
        self._interp: Dict[Any, Any] = {}
        self._opcache: Dict[str, torch.Tensor] = {}
        self._fp8_wbatch = None  # ops.QuantBatch over the weight operands (built with the operand table)
        self._fp8_conv_bf16: Dict[Any, bool] = {}  # fp8 mode: (weight key, M) -> this convolution stays on the bf16 3x3 kernel
        self._opkey = None
        self._op_ptr_key = None
        self._opbatch = None
        self._ws: Optional[torch.Tensor] = None
        self._luts: Dict[Tuple[bool, bool], torch.Tensor] = {}
        self._plans: Dict[str, Any] = {}
        self.buckets: List[GradBucket] = []
        self._bucket_of: Dict[int, Tuple[GradBucket, int]] = {}
        # called as hook(bucket, side_event) when a bucket's gradients are complete once the CURRENT stream and side_event
        # (a torch.cuda.Event on the weight-gradient side stream, or None) have been reached
        self.bucket_ready_hook: Optional[Callable[[GradBucket, Optional[torch.cuda.Event]], None]] = None
        self._build_buckets()

    # ------------------------------------------------------------------ parameters & buckets
    def head_modules(self):
        tr = getattr(self.rvfm, "translator", None)
        if tr is None:
            return []
        return [(t, tr.translator_heads[tr.legit_target_model_name_map[t]]) for t in tr.target_model_names]

    def _build_buckets(self) -> None:
        """Backward-completion order: heads (in forward order), then backbone layer groups from the top down."""
        self.buckets = []
        tr = getattr(self.rvfm, "translator", None)
        for t, hm in self.head_modules():
            pre = f"translator.translator_heads.{tr.legit_target_model_name_map[t]}."
            self.buckets.append(GradBucket(f"head:{t}", [(pre + n, p) for n, p in hm.named_parameters()]))
        vit = self.rvfm.backbone.model
        # three-layer groups from the top; the LAST layers to finish (2, 1, 0 + embeddings) are buckets of their own: what is still
        # to be exchanged when backward ends is one layer (28 MB of fp32 for DeiT-base) instead of three + embeddings (88 MB)
        groups = [(9, 12), (6, 9), (3, 6), (2, 3), (1, 2), (0, 1)]
        for gi, (lo, hi) in enumerate(groups):
            ps: List[Tuple[str, torch.nn.Parameter]] = []
            if gi == 0:
                ps += [("backbone.model.layernorm." + n, p) for n, p in vit.layernorm.named_parameters()]
            for li in range(hi - 1, lo - 1, -1):
                ps += [(f"backbone.model.layers.{li}." + n, p) for n, p in vit.layers[li].named_parameters()]
            if lo == 0:
                ps += [("backbone.model.embeddings." + n, p) for n, p in vit.embeddings.named_parameters()]
            self.buckets.append(GradBucket(f"vit:{lo}-{hi - 1}", ps))
        self._bucket_of = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._bucket_of[id(p)] = (b, i)

    def _cached_params(self, key: str, module: torch.nn.Module) -> List[torch.nn.Parameter]:
        """list(module.parameters()), walked once: the module tree walk cost 0.5 ms of host time per step (tools/host_profile.py).  The
        buckets built at construction already assume that the Parameter objects of the model stay the ones they were."""
        c = self.__dict__.setdefault("_param_lists", {})
        hit = c.get(key)
        if hit is None or hit[0] is not module:
            hit = c[key] = (module, list(module.parameters()))
        return hit[1]

    def all_params(self) -> List[torch.nn.Parameter]:
        return [p for b in self.buckets for p in b.params]

    def _grad(self, p: torch.nn.Parameter) -> Tuple[torch.Tensor, bool]:
        """(view of the flat bucket for p, accumulate?) and makes p.grad that view.  A frozen parameter (requires_grad False)
        keeps p.grad untouched: the kernel that produces its gradient together with a trainable neighbour's writes into the
        bucket slot as scratch, and the optimizer skips it."""
        b, i = self._bucket_of[id(p)]
        b.ensure(p.device)
        v = b.view(i)
        if not p.requires_grad:
            return v, False
        if p.grad is None:
            p.grad = v
            return v, False
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)  # foreign .grad tensor: adopt it (rare path, plain copy)
            p.grad = v
        return v, True

    def _bucket_done(self, b: GradBucket, side: Optional["_SideQueue"] = None) -> None:
        if self.bucket_ready_hook is not None:
            self.bucket_ready_hook(b, side.fence() if side is not None else None)
        if side is not None:
            side.join_at_backward_end()

    # ------------------------------------------------------------------ small helpers
    def _side_queue(self, device, nfloats: int) -> "_SideQueue":
        q = getattr(self, "_sideq", None)
        if q is None or q.device != device:
            q = _SideQueue(device, enabled=os.environ.get("THEIA_SIDE_STREAM", "1") != "0")
            self._sideq = q
        q.ensure_ws(nfloats)
        return q

    def ws(self, nfloats: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nfloats or self._ws.device != device:
            self._ws = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=device)
        return self._ws

    def _lut(self, do_rescale: bool, do_normalize: bool, device) -> torch.Tensor:
        k = (do_rescale, do_normalize)
        t = self._luts.get(k)
        if t is None or t.device != device:
            bb = self.rvfm.backbone
            t = torch.from_numpy(preprocess_lut(do_rescale, do_normalize, bb.image_mean, bb.image_std)).to(device)
            self._luts[k] = t
        return t

    def _plan(self, key: str, geo: Optional[Geometry] = None):
        geo = geo or self.geo224
        ck = key if key in ("conv16", "up31", "up64") else (key,) + geo.key
        if ck not in self._plans:
            C, D = self.D, self.D
            if key == "pad":  # reads the 14x14 patch tokens of z[b, ntok, C] in place: rows tok0 .. tok0+195 of every image
                self._plans[ck] = ops.plan_convT3x3(C, GRID, 1, 0, 0, in_bs=geo.ntok * C, in_off=geo.tok0 * C)
            elif key == "conv16":
                self._plans[ck] = ops.plan_conv3x3(C, 16)
            elif key == "up31":
                self._plans[ck] = ops.plan_convT3x3(C, 16, 2, 1, 0)
            elif key == "up64":
                self._plans[ck] = ops.plan_convT3x3(C, 31, 2, 0, 1)
            elif key == "patch":  # [b*P, 768] patch matrix -> token rows tok0 + p of h[b, ntok, D]
                self._plans[ck] = ops.rowmap([(0, 0, 0)], (geo.gh, geo.gw), (geo.gh, geo.gw), 1, 768, geo.P * 768, 0, geo.gw, 1, 0, 0,
                                             geo.ntok * D, geo.tok0 * D)
        return self._plans[ck]

    # ------------------------------------------------------------------ position embeddings on another patch grid
    def _interp_mats(self, geo: Geometry, device):
        """(W [P, 196], W^T zero-padded to [196, P4]) f32 on the device: the bicubic interpolation of the 14x14 position table
        to gh x gw as ONE matrix (the op is linear in the table).  DeiT: HF's F.interpolate(size=(gh, gw)) (transformers
        modeling_vit.py:89-127); nocls- / reg-: the reference's scale_factor = (g + 0.1) / 14 variant (backbones.py:39-69,146-177)."""
        k = (geo.gh, geo.gw, device)
        if k not in self._interp:
            own = self.tok0 == 0 or self.nreg > 0
            sy = 1.0 / ((geo.gh + 0.1) / GRID) if own else GRID / geo.gh
            sx = 1.0 / ((geo.gw + 0.1) / GRID) if own else GRID / geo.gw
            wy, wx = bicubic_matrix(GRID, geo.gh, sy), bicubic_matrix(GRID, geo.gw, sx)
            w = np.einsum("ai,bj->abij", wy, wx).reshape(geo.P, GRID * GRID).astype(np.float32)
            p4 = (geo.P + 3) // 4 * 4
            wt = np.zeros((GRID * GRID, p4), dtype=np.float32)
            wt[:, :geo.P] = w.T
            self._interp[k] = (torch.from_numpy(w).to(device), torch.from_numpy(wt).to(device), p4)
        return self._interp[k]

    def _patch_pos(self, geo: Geometry, device) -> torch.Tensor:
        """f32 [P, D]: the position rows added to the patch tokens"""
        pos = self.rvfm.backbone.model.embeddings.position_embeddings.view(GRID * GRID + 1, self.D)
        if not geo.interp:
            return pos[1:]
        w, _wt, _p4 = self._interp_mats(geo, device)
        posT = torch.empty(self.D, GRID * GRID, dtype=torch.float32, device=device)
        ops.cast_transpose(pos[1:], posT)
        return ops.linear(w, posT)  # exact-f32 MFMA path: [P, 196] @ [196, D]

    # ------------------------------------------------------------------ operand cache
    def _operands(self, device) -> Dict[str, torch.Tensor]:
        """bf16 / f32 GEMM operands (plain, transposed and packed copies of the fp32 master weights), rebuilt by ONE batched
        cast launch whenever a parameter changed (i.e. after every optimizer step)."""
        params = self.all_params()
        ptr_key = (device, self.dtype) + tuple((p.data_ptr(), tuple(p.shape)) for p in params)
        key = (PARAM_EPOCH[0],) + tuple(p._version for p in params)
        # A stream capture ALWAYS records the rebuild: a captured train step replays "rebuild, forward, backward, update", and if the
        # cache happened to be fresh at capture time (an eval forward between the last optimizer step and a re-capture) the replays
        # would run on frozen operand copies while AdamW keeps updating the masters (round-4 advisor finding)
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if ptr_key == self._op_ptr_key and key == self._opkey and not capturing:
            return self._opcache
        if ptr_key != self._op_ptr_key:
            self._opcache, self._opbatch = self._build_operand_table(device)
            self._op_ptr_key = ptr_key
            if self.precision == "fp8":
                self.fp8 = Fp8Scales(device)
                self._fp8_wbatch = None
        self._opbatch.run()
        if self.fp8 is not None:
            # delayed scaling: last step's maxima become this step's scales; then the e4m3 copies of every GEMM weight operand
            # (forward W and data-gradient W^T / packed convolution weights) are re-quantised from their bf16 operands
            self.fp8.update()
            if self._fp8_wbatch is None:
                # first build: every weight slot calibrates itself (max-only pass, then its scale) and gets its persistent e4m3 buffer;
                # from then on ONE launch re-quantises all of them (round 6: they were ~130 launches of a few us per step)
                qb = ops.QuantBatch(device)
                for k in [k for k, t in self._opcache.items() if t.dtype == torch.bfloat16 and t.dim() == 2 and not k.endswith(".f8")
                          and t.shape[1] % 64 == 0 and t.shape[0] >= 64 and k != "patch.w" and t.is_contiguous()]:
                    q, inv = self.fp8.quantize(self._opcache[k], "w:" + k)
                    self._opcache[k + ".f8"], self._opcache[k + ".inv"] = q, inv
                    i = self.fp8.slot("w:" + k)
                    qb.add(self._opcache[k], q, self.fp8.scale[i:i + 1], self.fp8.amax[i:i + 1])
                self._fp8_wbatch = qb
            else:
                self._fp8_wbatch.run()
        self._opkey = key
        return self._opcache

    def _q8_conv(self, wkey: str, like: torch.Tensor, kind: str):
        """_q8_for for the input of a convolution: only the kinds that never run on the bf16 3x3 kernel (the stride-2 transposed convolutions
        and the stride-1 pad's data-gradient), and inside the row range _conv_operands quantises"""
        if kind not in ("up31", "up64", "pad_d") or like.numel() // self.D >= (1 << 22):
            return None
        return self._q8_for(wkey, like)

    def _q8_for(self, key: str, like, records_max: bool = True):
        """the e4m3 side output a producer should write for the GEMM input of operand `key` (fp8 mode, site calibrated), else None"""
        if self.fp8 is None or key + ".f8" not in self._opcache:
            return None
        if isinstance(like, torch.Tensor) and (like.dtype != torch.bfloat16 or not like.is_contiguous()):
            return None
        if os.environ.get("THEIA_FP8_FUSED_QUANT", "1") == "0":  # A/B switch: every GEMM input through its own quantise pass
            return None
        return self.fp8.fused("x:" + key, like, records_max)

    def _mm(self, x: torch.Tensor, key: str, bias: Optional[torch.Tensor] = None, x8=None, **epi) -> torch.Tensor:
        """x [M, K] @ operand `key`^T with the epilogue `epi`: fp8 operands when the engine runs in fp8 mode and the operand has an
        e4m3 copy (K a multiple of 64, N >= 64), bf16 / f32 otherwise.  x8: what _q8_for(key, x) returned when x's producer already
        wrote the e4m3 copy."""
        oc = self._opcache
        use_fp8 = epi.pop("fp8", True)  # False: this launch keeps its bf16 operands in fp8 mode too (see the attention call sites)
        out8 = epi.pop("out8", None)  # _q8_for(<the consumer's key>, ((M, N), device)): this launch's output also as e4m3 -- fp8 launches only
        o8 = (out8[0], out8[1]) if out8 is not None else None
        if x8 is not None and x8[0] is None:
            x8 = None  # (its producer ran on bf16 operands and could not write it)
        if x8 is not None and "out" not in epi:
            return ops.linear(x8[0].view(x.shape), oc[key + ".f8"], bias, scale_inv=(x8[3], oc[key + ".inv"]), out8=o8, **epi)
        # (the fp8 operands exist for the persistent kernel only, which addresses M < 2^24 rows: beyond that -- a head's 64x64 maps at
        # b >= 4096 per GPU -- the launch keeps its bf16 operands instead of failing in the middle of a step)
        if use_fp8 and self.fp8 is not None and key + ".f8" in oc and "out" not in epi and x.shape[0] < (1 << 24):
            x8, inv = self.fp8.quantize(x, "x:" + key)
            return ops.linear(x8, oc[key + ".f8"], bias, scale_inv=(inv, oc[key + ".inv"]), out8=o8, **epi)
        if out8 is not None:
            out8[0] = None  # a bf16 launch: the consumer quantises its input itself
        return ops.linear(x, oc[key], bias, **epi)

    def _build_operand_table(self, device):
        T, D, F = self.dtype, self.D, self.F
        oc: Dict[str, torch.Tensor] = {}
        cb = ops.CastBatch(device, T)

        def new(name, *shape, dtype=T):
            t = torch.empty(*shape, dtype=dtype, device=device)
            oc[name] = t
            return t

        vit = self.rvfm.backbone.model
        cb.add_cast(vit.embeddings.patch_embeddings.projection.weight.view(D, 768), new("patch.w", D, 768))
        for i, L in enumerate(vit.layers):
            a = L.attention
            wqkv = new(f"l{i}.wqkv", 3 * D, D)
            wqkvT = new(f"l{i}.wqkvT", D, 3 * D)
            bqkv = new(f"l{i}.bqkv", 3 * D, dtype=torch.float32)
            for j, prj in enumerate((a.q_proj, a.k_proj, a.v_proj)):
                cb.add_cast(prj.weight, wqkv[j * D:(j + 1) * D])
                cb.add_transpose(prj.weight, wqkvT[:, j * D:], ldd=3 * D)
                cb.add_cast(prj.bias, bqkv[j * D:(j + 1) * D])
            cb.add_cast(a.o_proj.weight, new(f"l{i}.wo", D, D))
            cb.add_transpose(a.o_proj.weight, new(f"l{i}.woT", D, D))
            cb.add_cast(L.mlp.fc1.weight, new(f"l{i}.w1", F, D))
            cb.add_transpose(L.mlp.fc1.weight, new(f"l{i}.w1T", D, F))
            cb.add_cast(L.mlp.fc2.weight, new(f"l{i}.w2", D, F))
            cb.add_transpose(L.mlp.fc2.weight, new(f"l{i}.w2T", F, D))
        C = D
        for t, hm in self.head_modules():
            pf = f"h:{t}."
            if hm.kind == "cls":
                Ct = hm.adapter["0"].weight.shape[0]
                cb.add_cast(hm.adapter["0"].weight, new(pf + "w", Ct, C))
                cb.add_transpose(hm.adapter["0"].weight, new(pf + "wT", C, Ct))
                continue
            pad_plan = self._plan("pad")
            cb.add(hm.pad["1"].weight, new(pf + "pad.wf", C, 9 * C), *pad_plan.pack_fwd)
            cb.add(hm.pad["1"].weight, new(pf + "pad.wd", C, 9 * C), *pad_plan.pack_dgrad)
            for idx, pk in (("1", "up31"), ("4", "up64")) if hm.kind == "up64" else (("1", "conv16"), ("4", "conv16")):
                pl = self._plan(pk)
                cb.add(hm.adapter[idx].weight, new(pf + f"c{idx}.wf", C, 9 * C), *pl.pack_fwd)
                cb.add(hm.adapter[idx].weight, new(pf + f"c{idx}.wd", C, 9 * C), *pl.pack_dgrad)
            for idx, hw in zip(("0", "3", "6"), hm.sizes):
                HW = hw * hw
                # LN affine [C, H, W] -> [HW, C] (f32): a 2-D transpose, j = pixel (unit source stride), k = channel
                cb.add(hm.adapter[idx].weight, new(pf + f"ln{idx}.g", HW * C, dtype=torch.float32), 1, HW, C, 0, 1, HW)
                cb.add(hm.adapter[idx].bias, new(pf + f"ln{idx}.b", HW * C, dtype=torch.float32), 1, HW, C, 0, 1, HW)
            Ct = hm.adapter["8"].weight.shape[0]
            cb.add_cast(hm.adapter["8"].weight, new(pf + "w8", Ct, C))
            cb.add_transpose(hm.adapter["8"].weight, new(pf + "w8T", C, Ct))
        return oc, cb

    # ================================================================== backbone
    def backbone(self, x: Any, do_rescale: bool = True, do_normalize: bool = True, do_resize: bool = True,
                 interpolate_pos_encoding: bool = False) -> torch.Tensor:
        vit = self.rvfm.backbone.model
        device = vit.layernorm.weight.device
        if device.type != "cuda":
            raise RuntimeError("theia_amd runs on a ROCm GPU only: move the model with .to('cuda') (no CPU fallback)")
        batch = to_uint8_batch(x, any_size=True)
        img, channels_last = self._to_model_size(batch, device, do_resize, interpolate_pos_encoding)
        hh, ww = (img.shape[1], img.shape[2]) if channels_last else (img.shape[2], img.shape[3])
        gh, gw = hh // 16, ww // 16
        # HF interpolates unless the patch count matches the table and the image is square (modeling_vit.py:103-104)
        interp = interpolate_pos_encoding and not (gh * gw == GRID * GRID and hh == ww)
        geo = self.geo224 if (gh, gw, interp) == (GRID, GRID, False) else Geometry(gh, gw, self.tok0, self.nreg, interp)
        if geo.ntok > MAX_TOKENS:
            raise NotImplementedError(f"{hh}x{ww} input = {geo.ntok} tokens: the attention kernels hold at most {MAX_TOKENS}")
        params = self._cached_params("vit", vit)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _BackboneFn.apply(self, img, channels_last, do_rescale, do_normalize, geo, *params)
        if self.fp8 is not None:
            # delayed scaling outside training: the scales are otherwise refreshed only when the operand cache is rebuilt after an
            # optimizer step, so an inference / eval loop would keep the activation scales of its first batch for ever.  Only the
            # activation slots: a weight slot's scale belongs to the cached e4m3 copy of that weight
            self.fp8.update(activations_only=True)
        z, _ = self._backbone_fwd(img, channels_last, do_rescale, do_normalize, save=False, geo=geo)
        return z

    def _to_model_size(self, batch, device, do_resize: bool = True, any_final_size: bool = False) -> Tuple[torch.Tensor, bool]:
        """The processor's resize and center-crop steps (backbones.py:337-339; order resize -> crop as in the HF processor):
        theia_resize_u8 (Pillow's resampling, bit-exact) on the GPU to the configured size when do_resize, then the
        center crop (top = (H - crop) // 2).  Images already at the final size pass through untouched.  any_final_size:
        interpolate_pos_encoding was requested, the student runs at whatever size comes out."""
        bb = self.rvfm.backbone
        (rh, rw), crop, resample = bb.resize_size, bb.crop_size, bb.resample

        def one(t: torch.Tensor, cl: bool) -> torch.Tensor:
            t = t.to(device, non_blocking=True)
            hh, ww = (t.shape[1], t.shape[2]) if cl else (t.shape[2], t.shape[3])
            if do_resize and (hh, ww) != (rh, rw):
                t, cl, hh, ww = ops.resize_u8(t, cl, rh, rw, resample), True, rh, rw
            if crop is not None and (hh, ww) != (crop, crop):
                if hh < crop or ww < crop:
                    raise NotImplementedError(f"center-crop {crop} of a {hh}x{ww} image (the HF processor zero-pads) is not implemented")
                top, left = (hh - crop) // 2, (ww - crop) // 2
                t = t[:, top:top + crop, left:left + crop, :] if cl else t[:, :, top:top + crop, left:left + crop]
                hh = ww = crop
            if (hh, ww) != (IMAGE, IMAGE) and not any_final_size:  # HF ViTEmbeddings.forward raises the same way (modeling_vit.py:152-157)
                raise ValueError(f"Input image size ({hh}*{ww}) doesn't match model ({IMAGE}*{IMAGE}): keep do_resize=True or pass "
                                 f"interpolate_pos_encoding=True")
            if hh < 16 or ww < 16:
                raise ValueError(f"{hh}x{ww} image: smaller than one 16x16 patch")
            return t, cl

        if isinstance(batch, list):  # items of different sizes: one resize launch per item, then one batch
            outs = []
            for t, cl in batch:
                t, cl = one(t, cl)
                outs.append(t if cl else t.permute(0, 2, 3, 1))
            if len({tuple(o.shape[1:]) for o in outs}) > 1:
                raise ValueError("images of different sizes after the processor cannot be batched")
            return torch.cat(outs, 0).contiguous(), True
        img, cl = one(*batch)
        return img.contiguous(), cl

    def _backbone_fwd(self, img: torch.Tensor, channels_last: bool, do_rescale: bool, do_normalize: bool, save: bool,
                      geo: Optional[Geometry] = None):
        geo = geo or self.geo224
        dev, T, D, F, nh = img.device, self.dtype, self.D, self.F, self.heads
        NTOK, P = geo.ntok, geo.P
        oc = self._operands(dev)
        vit = self.rvfm.backbone.model
        emb = vit.embeddings
        b = img.shape[0]
        M = b * NTOK
        patches = torch.empty(b * P, 768, dtype=T, device=dev)
        ops.patchify(img, self._lut(do_rescale, do_normalize, dev), patches, channels_last)
        h = torch.empty(M, D, dtype=T, device=dev)
        pos = emb.position_embeddings.view(GRID * GRID + 1, D)
        ops.gemm_nt(patches, oc["patch.w"], h, b * P, D, 768, self._plan("patch", geo), 768, D,
                    bias=emb.patch_embeddings.projection.bias, rowtab=self._patch_pos(geo, dev), rowtab_period=P)
        if geo.tok0:  # CLS token + its position row (modeling_vit.py:148-149,159)
            ops.write_tokens(emb.cls_token.view(1, D), pos[:1], h, b, NTOK, 0, 1, D)
        if geo.nreg:  # register tokens + their own position rows, after the patches (backbones.py:196-205)
            ops.write_tokens(emb.reg_token.view(geo.nreg, D), emb.reg_pos_embed.view(geo.nreg, D), h, b, NTOK, geo.tok0 + P, geo.nreg, D)
        saved: Dict[str, Any] = {"b": b, "geo": geo, "patches": patches if save else None, "layers": []}
        for i, L in enumerate(vit.layers):
            q8a = self._q8_for(f"l{i}.wqkv", h)
            a, mean1, rstd1 = ops.layernorm_fwd(h, L.layernorm_before.weight, L.layernorm_before.bias, LN_EPS_VIT, q8=q8a[:3] if q8a is not None else None)
            qkv = self._mm(a, f"l{i}.wqkv", oc[f"l{i}.bqkv"], x8=q8a)
            o, lse = ops.attention_fwd(qkv, b, NTOK, nh)
            # (fp8 mode: the attention kernels write bf16 only, and a quantisation pass over o -- or over dQKV in backward -- costs more than
            # e4m3 operands save on these K = D / 3D launches: they keep their bf16 operands)
            h1 = self._mm(o, f"l{i}.wo", L.attention.o_proj.bias, resid=h, fp8=False)
            q8m = self._q8_for(f"l{i}.w1", h1)
            m, mean2, rstd2 = ops.layernorm_fwd(h1, L.layernorm_after.weight, L.layernorm_after.bias, LN_EPS_VIT, q8=q8m[:3] if q8m is not None else None)
            pre = torch.empty(M, F, dtype=T, device=dev) if save else None
            q8act = self._q8_for(f"l{i}.w2", ((M, F), dev), records_max=False)  # (fp8 mode: fc1's epilogue also writes the e4m3 operand of fc2)
            act = self._mm(m, f"l{i}.w1", L.mlp.fc1.bias, x8=q8m, act=N.ACT_GELU, aux_out=pre, out8=q8act)
            h2 = self._mm(act, f"l{i}.w2", L.mlp.fc2.bias, x8=q8act, resid=h1)
            if save:
                saved["layers"].append((h, mean1, rstd1, a, qkv, o, lse, h1, mean2, rstd2, m, pre, act))
            h = h2
        z, meanf, rstdf = ops.layernorm_fwd(h, vit.layernorm.weight, vit.layernorm.bias, LN_EPS_VIT)
        if save:
            saved["final"] = (h, meanf, rstdf)
        return z.view(b, NTOK, D), saved

    def _backbone_bwd(self, saved: Dict[str, Any], dz: torch.Tensor) -> None:
        dev, T, D, F, nh = dz.device, self.dtype, self.D, self.F, self.heads
        oc = self._operands(dev)
        vit = self.rvfm.backbone.model
        b, geo = saved["b"], saved["geo"]
        NTOK, P = geo.ntok, geo.P
        M = b * NTOK
        dz = dz.contiguous().view(M, D)
        if dz.dtype != T:
            raise TypeError(f"gradient dtype {dz.dtype} does not match the engine's compute dtype {T}")
        _wt = lambda n_, k_: max(1, N.lib().theia_wgrad_tiles(n_, k_))  # output tiles of a weight gradient (256 x 256 or 128 x 384)
        wsz = max(N.lib().theia_layernorm_bwd_workspace_bytes(M, D) // 4, N.lib().theia_colsum_workspace_bytes(M, F) // 4,
                  N.lib().theia_colsum_workspace_bytes(b, NTOK * D) // 4,
                  max(ops.wgrad_splits(M, n_, k_) * n_ * (k_ + 1) for n_, k_ in ((D, F), (F, D), (D, D), (3 * D, D), (D, 768))),  # slabs + bias partials
                  # the grouped q/k/v + o_proj launch: one split count for both problems
                  N.lib().theia_wgrad_group_splits(M, _wt(3 * D, D) + _wt(D, D)) * 4 * D * (D + 1),
                  # ... and with fc1 / fc2 in it (the small students)
                  N.lib().theia_wgrad_group_splits(M, _wt(3 * D, D) + _wt(D, D) + _wt(F, D) + _wt(D, F)) * (4 * D * (D + 1) + F * (D + 1) + D * (F + 1)))
        ws = self.ws(wsz, dev)
        side = self._side_queue(dev, wsz)

        # weight / bias gradients only feed the optimizer: they run on a side HIP stream so that their workgroups fill the
        # CUs left idle by the tail rounds and epilogues of the data-gradient chain on the main stream (and vice versa)
        def wgrad(dy, x, pw, pb):
            """weight and bias gradient of one nn.Linear: one GEMM launch (+ slab reduce) on the side stream"""
            if not (pw.requires_grad or pb.requires_grad):
                return
            gw_, accw_ = self._grad(pw)
            gb_, accb_ = self._grad(pb)
            side.run(lambda: ops.linear_wgrad(dy, x, gw_, accw_, side.ws, bias=(gb_, accb_)), dy, x)

        # Grouped launches (ops.linear_wgrad_group): a layer's o_proj gradient always waits for the fused q/k/v gradient and shares its
        # launch; when ALL FOUR weight gradients of a layer are few tiles (DeiT-small: 38, DeiT-tiny: 10 -- a launch of their own each
        # would be 4-64 M-splits of a handful of tiles, 30-70 us apiece), fc2's and fc1's wait as well: one launch per layer.  DeiT-base
        # (108 tiles = 2 splits on 216 of 256 CUs) keeps fc1 / fc2 on their own 36 x 7 launches.  THEIA_WGRAD_GROUP=0 / 1 / all: A/B.
        tl = lambda n_, k_: max(1, N.lib().theia_wgrad_tiles(n_, k_))
        gmode = os.environ.get("THEIA_WGRAD_GROUP", "auto")
        group_all = T == torch.bfloat16 and (gmode == "all" or (gmode == "auto" and tl(3 * D, D) + tl(D, D) + tl(F, D) + tl(D, F) <= 64))
        pending: List[Tuple[torch.Tensor, torch.Tensor, Any]] = []

        def wgrad_later(dy, x, lin):
            if group_all and lin.weight.requires_grad and lin.bias.requires_grad:
                pending.append((dy, x, lin))
            else:
                wgrad(dy, x, lin.weight, lin.bias)

        hL, meanf, rstdf = saved["final"]
        q8dh = self._q8_for(f"l{NUM_LAYERS - 1}.w2T", dz)  # (fp8 mode: the LayerNorm backward passes write the e4m3 copy their consumer GEMM reads)
        dh = self._ln_bwd(dz, hL, vit.layernorm, meanf, rstdf, None, ws, q8=q8dh)
        vit_buckets = [bk for bk in self.buckets if bk.name.startswith("vit:")]  # layer groups 9-11, 6-8, 3-5, then 2, 1, 0 (+ embeddings)
        group_lo = {int(bk.name[4:].split("-")[0]): bk for bk in vit_buckets[:-1]}  # lowest layer of a group -> its bucket
        for i in range(NUM_LAYERS - 1, -1, -1):
            L = vit.layers[i]
            (h, mean1, rstd1, a, qkv, o, lse, h1, mean2, rstd2, m, pre, act) = saved["layers"][i]
            saved["layers"][i] = None
            # h2 = h1 + fc2(act)
            wgrad_later(dh, act, L.mlp.fc2)
            q8dpre = self._q8_for(f"l{i}.w1T", ((M, F), dev), records_max=False)
            dpre = self._mm(dh, f"l{i}.w2T", None, x8=q8dh, act=N.ACT_MUL_DGELU, aux_in=pre, out8=q8dpre)
            del act, pre
            wgrad_later(dpre, m, L.mlp.fc1)
            dm = self._mm(dpre, f"l{i}.w1T", x8=q8dpre)
            del dpre
            q8dh1 = self._q8_for(f"l{i}.woT", dm)
            dh1 = self._ln_bwd(dm, h1, L.layernorm_after, mean2, rstd2, dh, ws, q8=q8dh1)
            del dm, dh
            # h1 = h + o_proj(o).  Its weight gradient ([D, D]: 9 output tiles, i.e. 28 M-splits alone) waits for the q/k/v gradient and
            # shares that launch (36 tiles x 7 splits; ops.linear_wgrad_group); alone only when the fused q/k/v path is not taken
            do = self._mm(dh1, f"l{i}.woT", x8=q8dh1)
            dqkv = ops.attention_bwd(qkv, o, do, lse, b, NTOK, nh, ws)
            del do
            qkv_mods = (L.attention.q_proj, L.attention.k_proj, L.attention.v_proj)
            also = [(dh1, o, L.attention.o_proj)] + pending
            pending = []
            if not self._wgrad_qkv_fused(dqkv, a, qkv_mods, side, also=also):
                for dy_, x_, lin_ in also:
                    wgrad(dy_, x_, lin_.weight, lin_.bias)
                for j, prj in enumerate(qkv_mods):
                    sl = dqkv[:, j * D:(j + 1) * D]
                    wgrad(sl, a, prj.weight, prj.bias)
            da = self._mm(dqkv, f"l{i}.wqkvT", fp8=False)
            del dqkv
            q8dh = self._q8_for(f"l{i - 1}.w2T", da) if i > 0 else None
            dh = self._ln_bwd(da, h, L.layernorm_before, mean1, rstd1, dh1, ws, q8=q8dh)
            del da, dh1
            if i in group_lo:
                self._bucket_done(group_lo[i], side)
        # embeddings: h0[b, 0] = cls + pos[0];  h0[b, tok0+p] = patches @ Wp^T + bias + ppos[p];  h0[b, tok0+P+r] = reg[r] + reg_pos[r]
        emb = vit.embeddings
        gpos, acc = self._grad(emb.position_embeddings)
        tmp = None
        if geo.key == (GRID, GRID, 1, 0, False):  # plain DeiT at 224: the token sums ARE the position-embedding gradient
            ops.colsum(dh.view(b, NTOK * D), gpos.view(NTOK * D), acc, ws)
        else:
            tmp = torch.empty(NTOK * D, dtype=torch.float32, device=dev)  # sum over the batch, per token
            ops.colsum(dh.view(b, NTOK * D), tmp, False, ws)
            gp = gpos.view(GRID * GRID + 1, D)
            if geo.tok0:
                ops.unpermute3(tmp[:D], gp[0], 1, 1, D, 0, 0, 1, acc)
            elif not acc:
                ops.fill_zero(gp[0])  # nocls-: position row 0 is never used (backbones.py:91)
            tp = tmp[geo.tok0 * D:(geo.tok0 + P) * D]
            if not geo.interp:
                ops.unpermute3(tp, gp[1:], 1, 1, P * D, 0, 0, 1, acc)
            else:  # through the interpolation matrix: d pos[1:] = W^T @ d ppos   (exact-f32 MFMA path)
                _w, wt, p4 = self._interp_mats(geo, dev)
                tpT = torch.zeros(D, p4, dtype=torch.float32, device=dev)
                ops.cast_transpose(tp.view(P, D), tpT, ldd=p4)
                ops.linear(wt, tpT, out=gp[1:], resid=gp[1:] if acc else None)
            if geo.nreg:
                treg = tmp[(geo.tok0 + P) * D:]
                g, a2 = self._grad(emb.reg_token)
                ops.unpermute3(treg, g.view(-1), 1, 1, geo.nreg * D, 0, 0, 1, a2)
                g, a2 = self._grad(emb.reg_pos_embed)
                ops.unpermute3(treg, g.view(-1), 1, 1, geo.nreg * D, 0, 0, 1, a2)
        if geo.tok0:
            gcls, acc = self._grad(emb.cls_token)
            ops.colsum(dh.view(b, NTOK * D)[:, :D], gcls.view(D), acc, ws)
        gpb, acc = self._grad(emb.patch_embeddings.projection.bias)
        # bias gradient = sum over images and patch tokens: two column sums (tokens of one image, then images)
        if tmp is None:
            tmp = torch.empty(NTOK * D, dtype=torch.float32, device=dev)
            ops.colsum(dh.view(b, NTOK * D), tmp, False, ws)
        ops.colsum(tmp.view(NTOK, D)[geo.tok0:geo.tok0 + P], gpb, acc, ws)
        gpw, acc = self._grad(emb.patch_embeddings.projection.weight)
        rmap = self._plan("patch", geo)
        Mp = b * P
        splits = ops.wgrad_splits(Mp, D, 768)
        slabs = ws[: splits * D * 768]
        ops.gemm_wgrad(dh, saved["patches"], slabs, Mp, D, D, 1, splits, rmap)
        ops.wgrad_finish(slabs, splits, D, 1, 768, gpw, 768, 0, 1, acc)
        self._bucket_done(vit_buckets[-1], side)

    def _ln_bwd(self, dy, x, ln, mean, rstd, dresid, ws, q8=None):
        """row-LayerNorm backward with the affine gradients written to the bucket.  The kernel takes ONE accumulate flag for weight and
        bias; when the two parameters disagree (one frozen, or only one of the .grad tensors reset to None under gradient
        accumulation) the slot that must NOT accumulate is zeroed first and the kernel accumulates into both."""
        gw, accw = self._grad(ln.weight)
        gb, accb = self._grad(ln.bias)
        if accw != accb:
            ops.fill_zero(gw if not accw else gb)
            accw = accb = True
        return ops.layernorm_bwd(dy, x, ln.weight, mean, rstd, dresid, gw, gb, accw, ws, q8=q8[:3] if q8 is not None else None)

    def _wgrad_qkv_fused(self, dqkv: torch.Tensor, a: torch.Tensor, mods, side: "_SideQueue", also=None) -> bool:
        """The q / k / v weight and bias gradients as ONE [3D, D] weight-gradient GEMM when their slots in the flat gradient bucket
        are adjacent (they are: consecutive parameters of one layer, D*D and D multiples of 8): 27 output tiles x 9 row splits
        instead of 3 x (9 tiles x 28 splits) -- a third of the f32 slab traffic (64 MB instead of 198 MB per layer, written and
        read back) and a third of the launches.  Returns False (caller falls back to three GEMMs) when a slot is frozen or the
        slots are not adjacent.  also = [(dy, x, linear)]: more nn.Linear layers with the same M whose gradients join the launch (round 6:
        o_proj, for the small students fc2 and fc1 too -- ops.linear_wgrad_group; THEIA_WGRAD_GROUP=0: A/B switch, one launch each)."""
        D = self.D
        ws_, bs_ = [m.weight for m in mods], [m.bias for m in mods]
        if not all(p.requires_grad for p in ws_ + bs_) or os.environ.get("THEIA_QKV_WGRAD") == "split":
            return False
        # the bail-out checks must not touch p.grad (self._grad makes it the bucket view: a fallback that called it again would see
        # "gradient present" and accumulate onto stale bucket contents): inspect the bucket slots and the .grad state directly
        def slot(p):
            b, i = self._bucket_of[id(p)]
            return b, b.offsets[i]

        def will_accumulate(p):
            return p.grad is not None

        if len({will_accumulate(p) for p in ws_ + bs_}) != 1:
            return False
        for lst, n in ((ws_, D * D), (bs_, D)):
            for p0, p1 in zip(lst, lst[1:]):
                (b0, o0), (b1, o1) = slot(p0), slot(p1)
                if b0 is not b1 or o1 != o0 + n:
                    return False
            for p in lst:  # a foreign .grad tensor (not the bucket view) would need the copy self._grad does: take the slow path
                if p.grad is not None:
                    b, o = slot(p)
                    if b.flat is None or p.grad.data_ptr() != b.flat.data_ptr() + 4 * o:
                        return False
        gw = [self._grad(p) for p in ws_]
        gb = [self._grad(p) for p in bs_]
        acc = gw[0][1]
        gw_all = torch.as_strided(gw[0][0], (3 * D, D), (D, 1))   # views over the three adjacent bucket slots
        gb_all = torch.as_strided(gb[0][0], (3 * D,), (1,))
        probs = [(dqkv, a, gw_all, acc, (gb_all, acc))]
        for dy2, x2, lin in (also or []):
            if not (lin.weight.requires_grad or lin.bias.requires_grad):
                continue
            gw2, accw2 = self._grad(lin.weight)
            gb2, accb2 = self._grad(lin.bias)
            if lin.weight.requires_grad and lin.bias.requires_grad and os.environ.get("THEIA_WGRAD_GROUP", "auto") != "0":
                probs.append((dy2, x2, gw2, accw2, (gb2, accb2)))
            else:
                side.run(lambda dy2=dy2, x2=x2, gw2=gw2, accw2=accw2, gb2=gb2, accb2=accb2:
                         ops.linear_wgrad(dy2, x2, gw2, accw2, side.ws, bias=(gb2, accb2)), dy2, x2)
        if len(probs) > 1:
            side.run(lambda: ops.linear_wgrad_group(probs, side.ws), *[t for pr in probs for t in pr[:2]])
        else:
            side.run(lambda: ops.linear_wgrad(dqkv, a, gw_all, acc, side.ws, bias=(gb_all, acc)), dqkv, a)
        return True

    # ================================================================== translator heads
    def translator(self, z: torch.Tensor, names: List[str]) -> Dict[str, torch.Tensor]:
        tr = self.rvfm.translator
        for t in names:
            if t not in tr.legit_target_model_name_map:
                raise KeyError(t)
        head_params: List[torch.nn.Parameter] = []
        for t in names:
            head_params += self._cached_params("head:" + t, tr.translator_heads[tr.legit_target_model_name_map[t]])
        if z.dtype != self.dtype:
            raise TypeError(f"feature dtype {z.dtype} does not match the engine's compute dtype {self.dtype}")
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in head_params)):
            outs = _TranslatorFn.apply(self, names, z, *head_params)
        else:
            outs, _ = self._translator_fwd(z, names, save=False)
        return {t: o for t, o in zip(names, outs)}

    def _head(self, t: str):
        tr = self.rvfm.translator
        return tr.translator_heads[tr.legit_target_model_name_map[t]]

    def _conv_fwd(self, x, wf, bias, plan, b, out, relu: bool, sums: Optional[torch.Tensor] = None, x8=None):
        """sums: zeroed f32 [b, 2]; every launch (4 output-parity classes for a stride-2 transposed convolution) adds the per-sample
        (sum, sum of squares) of what it stores: the statistics of the whole-sample LayerNorm that follows."""
        C = self.D
        wf, scale_inv = self._conv_operands(x, wf, None if len(plan.fwd) > 1 else (plan.fwd[0][0], b * plan.fwd[0][1], out), x8=x8)
        if scale_inv is not None:
            x = scale_inv[2]
        for rmap, mpi in plan.fwd:
            ops.gemm_nt(x, wf, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, bias=bias, act=N.ACT_RELU if relu else N.ACT_NONE,
                        ln_sums=sums, scale_inv=scale_inv[:2] if scale_inv is not None else None)
        return out

    def _conv_operands(self, x: torch.Tensor, wkey: str, launch=None, x8=None):
        """(weight operand, None) -- or in fp8 mode (e4m3 weight, (inv_x, inv_w, e4m3 activation)): the activation (any NHWC /
        token layout with C channels innermost) is quantised as a [rows, C] matrix, the row map addresses it unchanged.
        launch = (row map, M, out) of a single-launch convolution: when the bf16 launch would run on the one-image-per-tile 3x3 kernel
        (gemm_conv_pp: 1.25-1.5 PF, faster than the generic NT kernel is on e4m3 operands) it keeps its bf16 operands in fp8 mode too --
        no quantisation pass for its input either (round 6)."""
        oc = self._opcache
        if self.fp8 is not None and launch is not None:
            rmap, M, out = launch
            key = (wkey, M)
            keep = self._fp8_conv_bf16.get(key)
            if keep is None:
                C = self.D
                keep = self._fp8_conv_bf16[key] = ops.gemm_nt(x, oc[wkey], out, M, C, rmap.ntaps * C, rmap, 9 * C, C, plan_only=True) == 256009
            if keep:
                return oc[wkey], None
        if x8 is not None:  # (the producer of x wrote the e4m3 copy: _q8_conv)
            return oc[wkey + ".f8"], (x8[3], oc[wkey + ".inv"], x8[0].view(-1, self.D))
        if self.fp8 is not None and wkey + ".f8" in oc and x.numel() // self.D < (1 << 22):  # (output rows <= 4x input rows < 2^24: see _mm)
            x8, inv = self.fp8.quantize(x.reshape(-1, self.D), "x:" + wkey)
            return oc[wkey + ".f8"], (inv, oc[wkey + ".inv"], x8)
        return oc[wkey], None

    def _translator_fwd(self, z: torch.Tensor, names: List[str], save: bool):
        dev, T, C = z.device, self.dtype, self.D
        oc = self._operands(dev)
        b, NTOK = z.shape[0], z.shape[1]
        if NTOK != self.geo224.ntok:
            raise NotImplementedError(f"the translator heads take the 14x14 patch grid of a 224x224 input ({self.geo224.ntok} tokens), "
                                      f"got {NTOK} tokens (the reference's heads reshape to 14x14 too, adapter_heads.py:281)")
        z = z.contiguous()
        outs, saved = [], []
        for t in names:
            hm = self._head(t)
            pf = f"h:{t}."
            if hm.kind == "cls":  # Linear on token 0: rows of z with stride NTOK*C (adapter_heads.py:50-57)
                if not self.tok0:
                    raise AssertionError("LinearAdapterHead needs a CLS token (adapter_heads.py:53: assert backbone_no_cls == False)")
                outs.append(ops.linear(z[:, 0, :], oc[pf + "w"], hm.adapter["0"].bias))
                if save:
                    saved.append(None)
                continue
            s0, s1, s2 = hm.sizes
            chw_ws = self.ws(N.lib().theia_layernorm_chw_workspace_bytes(b, s2 * s2 * C) // 4, dev)
            sums = torch.zeros(3, b, 2, dtype=torch.int64, device=dev)  # LayerNorm statistics out of the convolutions' epilogues (fixed point)
            u1 = torch.empty(b, 256 * C, dtype=T, device=dev)
            self._conv_fwd(z, pf + "pad.wf", hm.pad["1"].bias, self._plan("pad"), b, u1, relu=False, sums=sums[0])
            p1, p4 = ("up31", "up64") if hm.kind == "up64" else ("conv16", "conv16")
            # (fp8 mode: each LayerNorm apply also writes the e4m3 copy its consumer GEMM reads, where that consumer takes e4m3 operands)
            q1 = self._q8_conv(pf + "c1.wf", u1, p1)
            v1, st0 = ops.layernorm_chw_fwd(u1, oc[pf + "ln0.g"], oc[pf + "ln0.b"], LN_EPS_HEAD, chw_ws, sums=sums[0], q8=q1[:3] if q1 else None)
            u2 = torch.empty(b, s1 * s1 * C, dtype=T, device=dev)
            self._conv_fwd(v1, pf + "c1.wf", hm.adapter["1"].bias, self._plan(p1), b, u2, relu=True, sums=sums[1], x8=q1)
            q2 = self._q8_conv(pf + "c4.wf", u2, p4)
            v2, st3 = ops.layernorm_chw_fwd(u2, oc[pf + "ln3.g"], oc[pf + "ln3.b"], LN_EPS_HEAD, chw_ws, sums=sums[1], q8=q2[:3] if q2 else None)
            u3 = torch.empty(b, s2 * s2 * C, dtype=T, device=dev)
            self._conv_fwd(v2, pf + "c4.wf", hm.adapter["4"].bias, self._plan(p4), b, u3, relu=True, sums=sums[2], x8=q2)
            q3 = self._q8_for(pf + "w8", u3) if b * s2 * s2 < (1 << 24) else None
            v3, st6 = ops.layernorm_chw_fwd(u3, oc[pf + "ln6.g"], oc[pf + "ln6.b"], LN_EPS_HEAD, chw_ws, sums=sums[2], q8=q3[:3] if q3 else None)
            pred = self._mm(v3.view(b * s2 * s2, C), pf + "w8", hm.adapter["8"].bias, x8=q3)
            outs.append(pred.view(b, s2 * s2, -1))
            if save:
                saved.append((u1, st0, v1, u2, st3, v2, u3, st6, v3))
        return tuple(outs), {"z": z if save else None, "heads": saved, "b": b}

    def _translator_bwd(self, saved, names: List[str], dpreds: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
        z = saved["z"]
        dev, T, C, b = z.device, self.dtype, self.D, saved["b"]
        oc = self._operands(dev)
        dz = torch.zeros(b, self.geo224.ntok, C, dtype=T, device=dev)  # register-token rows stay zero (stripped before the heads)
        for hi, t in enumerate(names):
            dp = dpreds[hi]
            hm = self._head(t)
            pf = f"h:{t}."
            if hm.kind == "cls":
                if dp is None:
                    continue
                lin = hm.adapter["0"]
                Ct = lin.weight.shape[0]
                dp = dp.contiguous().view(b, Ct)
                if dp.dtype != T:
                    raise TypeError(f"gradient dtype {dp.dtype} does not match the engine's compute dtype {T}")
                z0, dz0 = z[:, 0, :], dz[:, 0, :]
                if lin.weight.requires_grad:
                    need = ops.wgrad_splits(b, Ct, C) * Ct * (C + 1)
                    side = self._side_queue(dev, need)
                    gw, accw = self._grad(lin.weight)
                    gb, accb = self._grad(lin.bias)
                    side.run(lambda dp=dp, z0=z0, gw=gw, accw=accw, gb=gb, accb=accb:
                             ops.linear_wgrad(dp, z0, gw, accw, side.ws, bias=(gb, accb)), dp, z)
                    ops.linear(dp, oc[pf + "wT"], out=dz0, resid=dz0)  # dz[:, 0] += dp @ W (several cls heads accumulate)
                    self._bucket_done(self._bucket_of[id(lin.weight)][0], side)
                else:
                    ops.linear(dp, oc[pf + "wT"], out=dz0, resid=dz0)
                continue
            bucket = self._bucket_of[id(hm.adapter["8"].weight)][0]
            if dp is None:
                continue
            train = hm.adapter["8"].weight.requires_grad
            (u1, st0, v1, u2, st3, v2, u3, st6, v3) = saved["heads"][hi]
            saved["heads"][hi] = None
            s0, s1, s2 = hm.sizes
            Ct = dp.shape[-1]
            E3 = s2 * s2 * C
            chw_need = N.lib().theia_layernorm_chw_workspace_bytes(b, E3) // 4
            conv_slabs = max(ops.conv_wgrad_splits(self._plan(k), b, C) for k in ("pad", "conv16", "up31", "up64")) * (9 * C * C + C)
            lin_slabs = ops.wgrad_splits(b * s2 * s2, Ct, C) * Ct * (C + 1)
            ws = self.ws(max(chw_need + 2 * E3 + 64, conv_slabs, lin_slabs,
                             N.lib().theia_colsum_workspace_bytes(b * s2 * s2, max(C, Ct)) // 4 + 64), dev)
            dp = dp.contiguous().view(b * s2 * s2, Ct)
            if dp.dtype != T:
                raise TypeError(f"gradient dtype {dp.dtype} does not match the engine's compute dtype {T}")

            side = self._side_queue(dev, max(conv_slabs, lin_slabs, N.lib().theia_colsum_workspace_bytes(b * s2 * s2, max(C, Ct)) // 4 + 64))

            def lin_grads(dy, x, mod):
                if not train:
                    return
                gw, accw = self._grad(mod.weight)
                gb, accb = self._grad(mod.bias)

                side.run(lambda: ops.linear_wgrad(dy, x, gw, accw, side.ws, bias=(gb, accb)), dy, x)

            def conv_grads(dy2d, x, mod, plan, mtot, bias_done=False):
                """dy2d [M_total, C] is the conv output gradient, x the conv input (flat NHWC); side stream like the ViT's.
                bias_done: the bias gradient came out of the LayerNorm backward that produced dy2d (ln_bwd(bias_of=))."""
                if not train:
                    return
                bias = None if bias_done else self._grad(mod.bias)
                gw, accw = self._grad(mod.weight)

                side.run(lambda: ops.conv_wgrad(plan, dy2d, x, b, C, gw, accw, side.ws, bias=bias), dy2d, x)

            def ln_bwd(dy, x, stats, idx, hw, relu_mask, bias_of=None, q8=None):
                """bias_of: the convolution that produced x, when its weight-gradient GEMM cannot carry the bias gradient (the stride-2
                transposed convolutions reduce over input pixels): its bias gradient = the column sums of dx, out of this pass."""
                E = hw * hw * C
                tmp_g = ws[chw_need: chw_need + E]
                tmp_b = ws[chw_need + E3: chw_need + E3 + E]
                dxsum = self._grad(bias_of.bias) if (train and bias_of is not None) else None
                dx = ops.layernorm_chw_bwd(dy, x, oc[pf + f"ln{idx}.g"], stats, tmp_g, tmp_b, relu_mask, False, ws[:chw_need], dxsum=dxsum,
                                           q8=q8[:3] if q8 else None)
                if train:
                    g, acc = self._grad(hm.adapter[idx].weight)
                    gb_, accb_ = self._grad(hm.adapter[idx].bias)
                    ops.transpose_acc2(tmp_g, g, acc, tmp_b, gb_, accb_, hw * hw, C)  # [HW][C] (NHWC reduction order) -> [C][H][W]
                return dx

            def conv_dgrad(dy, wd_key, plan, out, resid=None, x8=None):
                rmap, mpi = plan.dgrad
                wd, scale_inv = self._conv_operands(dy, wd_key, (rmap, b * mpi, out), x8=x8)
                ops.gemm_nt(dy if scale_inv is None else scale_inv[2], wd, out, b * mpi, C, 9 * C, rmap, 9 * C, C, resid=resid,
                            scale_inv=scale_inv[:2] if scale_inv is not None else None)
                return out

            p1, p4 = ("up31", "up64") if hm.kind == "up64" else ("conv16", "conv16")
            q8dp = getattr(dpreds[hi], "_theia_q8", None)
            q8dp = q8dp[1] if q8dp is not None and q8dp[0] == dp.data_ptr() and q8dp[1][0].numel() == dp.numel() else None
            lin_grads(dp, v3.view(b * s2 * s2, C), hm.adapter["8"])
            dv3 = self._mm(dp, pf + "w8T", x8=q8dp)
            del v3
            swapped4, swapped1 = self._plan(p4).wgrad_swapped, self._plan(p1).wgrad_swapped
            q6 = self._q8_conv(pf + "c4.wd", u3, p4)
            du3 = ln_bwd(dv3.view(b, E3), u3, st6, "6", s2, True, bias_of=hm.adapter["4"] if swapped4 else None, q8=q6)
            del dv3, u3
            conv_grads(du3.view(b * s2 * s2, C), v2, hm.adapter["4"], self._plan(p4), b * s2 * s2, bias_done=swapped4)
            dv2 = conv_dgrad(du3, pf + "c4.wd", self._plan(p4), torch.empty(b, s1 * s1 * C, dtype=T, device=dev), x8=q6)
            del du3, v2
            q3b = self._q8_conv(pf + "c1.wd", u2, p1)
            du2 = ln_bwd(dv2, u2, st3, "3", s1, True, bias_of=hm.adapter["1"] if swapped1 else None, q8=q3b)
            del dv2, u2
            conv_grads(du2.view(b * s1 * s1, C), v1, hm.adapter["1"], self._plan(p1), b * s1 * s1, bias_done=swapped1)
            dv1 = conv_dgrad(du2, pf + "c1.wd", self._plan(p1), torch.empty(b, 256 * C, dtype=T, device=dev), x8=q3b)
            del du2, v1
            q0 = self._q8_conv(pf + "pad.wd", u1, "pad_d") if self._fp8_conv_bf16.get((pf + "pad.wd", b * self._plan("pad").dgrad[1])) is False else None
            du1 = ln_bwd(dv1, u1, st0, "0", s0, False, q8=q0)
            del dv1, u1
            conv_grads(du1.view(b * 256, C), z, hm.pad["1"], self._plan("pad"), b * 256)
            conv_dgrad(du1, pf + "pad.wd", self._plan("pad"), dz, resid=dz, x8=q0)
            del du1
            if train:
                self._bucket_done(bucket, side)
        return dz

    # ================================================================== loss
    def distill_loss(self, pred: torch.Tensor, target: torch.Tensor, head: Optional[str] = None) -> torch.Tensor:
        """-> f32[3] = (mse, cos, smooth_l1) of one teacher (models/rvfm.py:153-168).  head: the teacher whose translator head produced
        `pred` (fp8 mode: the gradient pass then also writes the e4m3 copy that head's Linear data-gradient GEMM reads)."""
        # float32 as the reference feeds them (.float() of bf16-normalised features, train_rvfm.py:112-114, data_utils.py:374-379) -- or
        # those same values still in bf16 beside bf16 predictions: identical losses and gradients, 2 bytes less per element and pass
        if target.dtype != torch.float32 and not (target.dtype == torch.bfloat16 and pred.dtype == torch.bfloat16):
            raise TypeError("teacher features must be float32 (the reference feeds .float() tensors, train_rvfm.py:112-114), "
                            "or bfloat16 for a bf16 / fp8 model")
        if pred.shape != target.shape:
            raise ValueError(f"prediction {tuple(pred.shape)} and target {tuple(target.shape)} shapes differ")
        target = target.to(pred.device).contiguous()
        if torch.is_grad_enabled() and pred.requires_grad:
            return _LossFn.apply(self, pred, target, head)
        b = pred.shape[0]
        losses, _ = ops.distill_loss_fwd(pred.contiguous().view(b, -1), target.view(b, -1),
                                         self.ws(N.lib().theia_distill_loss_workspace_bytes(b, pred[0].numel()) // 4, pred.device))
        return losses


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: StudentEngine, img, channels_last, do_rescale, do_normalize, geo, *params):
        z, saved = eng._backbone_fwd(img, channels_last, do_rescale, do_normalize, save=True, geo=geo)
        ctx.eng, ctx.saved, ctx.nparams = eng, saved, len(params)
        return z

    @staticmethod
    def backward(ctx, dz):
        ctx.eng._backbone_bwd(ctx.saved, dz)
        ctx.saved = None
        return (None,) * (6 + ctx.nparams)


class _TranslatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: StudentEngine, names, z, *params):
        outs, saved = eng._translator_fwd(z, names, save=True)
        ctx.eng, ctx.saved, ctx.names, ctx.nparams = eng, saved, names, len(params)
        ctx.set_materialize_grads(False)  # heads whose prediction is unused get dpred = None and are skipped
        return outs

    @staticmethod
    def backward(ctx, *dpreds):
        dz = ctx.eng._translator_bwd(ctx.saved, ctx.names, dpreds)
        ctx.saved = None
        return (None, None, dz) + (None,) * ctx.nparams


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: StudentEngine, pred, target, head=None):
        ctx.eng, ctx.head = eng, head
        b = pred.shape[0]
        p2 = pred.contiguous().view(b, -1)
        t2 = target.view(b, -1)
        losses, coef = ops.distill_loss_fwd(p2, t2, eng.ws(N.lib().theia_distill_loss_workspace_bytes(b, p2.shape[1]) // 4, pred.device))
        ctx.p2, ctx.t2, ctx.coef, ctx.shape = p2, t2, coef, pred.shape
        return losses

    @staticmethod
    def backward(ctx, dl):
        q8 = ctx.eng._q8_for(f"h:{ctx.head}.w8T", ctx.p2) if ctx.head is not None and ctx.p2.shape[0] * (ctx.p2.shape[1] // ctx.shape[-1]) < (1 << 24) else None
        dp = ops.distill_loss_bwd(ctx.p2, ctx.t2, ctx.coef, dl.contiguous().float(), q8=q8[:3] if q8 else None)
        out = dp.view(ctx.shape)
        if q8 is not None:  # travels to _translator_bwd on the gradient tensor itself (checked there against its data pointer)
            out._theia_q8 = (dp.data_ptr(), q8)
        return None, out, None, None
