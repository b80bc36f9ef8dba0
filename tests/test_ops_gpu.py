"""GPU parity tests of every HIP kernel family, through the C ABI, against the CPU oracle / committed goldens.

f32 results must match the oracle to <= 1e-4 relative (the north-star tolerance); bf16 results are compared with
an f32 evaluation of the SAME bf16-rounded inputs at a bf16-appropriate tolerance (stated per test).
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import theia_oracle as O  # noqa: E402  (checker only)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def h(shape, seed, scale=1.0):
    return torch.from_numpy((O._hash_uniform(int(np.prod(shape)), seed) * scale).reshape(shape).copy())


def relerr(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def elem_excess(out, ref, rel=2.0 ** -8):
    """Elementwise criterion (relerr above is relative to the LARGEST entry, blind to the small ones): the largest amount by which an
    entry's error exceeds `rel` times that entry's own magnitude.  A bf16 result computed with f32 accumulation is the correctly rounded
    f32 value up to the accumulation-order noise, so this must stay at that noise (~1e-5 for O(1) data) however small the entry."""
    out = out.double().cpu()
    ref = ref.double().cpu()
    return float(((out - ref).abs() - rel * ref.abs()).max())


ELEM_FLOOR = 1e-4  # absolute: f32 accumulation-order differences + the bf16 GELU polynomial (3e-5), for O(1) operands


def rnd(x, dt):
    """value after rounding to the compute dtype (as f32 CPU tensor)."""
    return x.to(dt).float()


# ------------------------------------------------------------------------------------------------
def test_library_loads_and_abi():
    from theia_amd import _native as N
    lib = N.lib()
    assert lib.theia_abi_version() == N.ABI_VERSION == 12
    assert lib.theia_dtype_size(N.BF16) == 2


def test_probe_tr16_semantics():
    """ds_read_b64_tr_b16: lanes 4r..4r+3 of a 16-lane group supply row r; lane i receives column i, element j = row j."""
    from theia_amd import ops
    dev = _dev()
    img = torch.arange(1024, dtype=torch.int16)
    addr = torch.zeros(64, dtype=torch.int32)
    # each 16-lane group g reads a 4x16 block whose rows are 64 B apart (row pitch 64 B), group base g*256 B
    for l in range(64):
        g, q = l >> 4, l & 15
        addr[l] = g * 256 + (q >> 2) * 64 + (q & 3) * 8
    out = ops.probe_tr16(img.to(dev), addr.to(dev)).cpu().view(64, 4)
    for l in range(64):
        g, q = l >> 4, l & 15
        for j in range(4):
            expect = (g * 256 + j * 64) // 2 + q
            assert int(out[l, j]) == expect, (l, j, int(out[l, j]), expect)


@pytest.mark.parametrize("tile", [0, 128128, 256256])
def test_bf16_gelu_epilogue_over_a_wide_pre_activation_range(tile):
    """The bf16 GELU epilogue evaluates Phi as a clamped polynomial (gemm_tile.h: gt_phi_sat).  Pre-activations swept over [-50, 50]
    (an identity-like GEMM: x = the sweep in column block k, W = one-hot rows): the output must be EXACTLY 0 for x <= -4.5 (never a
    small value of either sign), exactly bf16(x) for x >= 4.5, and within 2e-4 absolute of erf-GELU in between; the GELU-gradient
    epilogue stays within [0 - 0.13, 1 + 0.13] (the exact derivative's range) and within 1e-3 of the exact derivative."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    M, K, N = 4096, 64, 256
    xs = torch.linspace(-50.0, 50.0, M).to(torch.bfloat16).float()
    xs[:512] = torch.linspace(-6.0, 6.0, 512).to(torch.bfloat16).float()  # dense around the interesting part
    x = torch.zeros(M, K)
    x[:, 0] = xs
    w = torch.zeros(N, K)
    w[:, 0] = 1.0  # every output column = x[:, 0]
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    y = ops.linear(x.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16), None, act=Nn.ACT_GELU, aux_out=pre, tile=tile).float().cpu()
    assert torch.equal(pre.float().cpu(), xs[:, None].expand(M, N))
    lo, hi = xs <= -4.5, xs >= 4.5
    assert float(y[lo].abs().max()) == 0.0
    assert torch.equal(y[hi], xs[hi, None].expand(-1, N))
    ref = torch.nn.functional.gelu(xs.double())[:, None]
    mid = ~(lo | hi)
    assert float(((y[mid].double() - ref[mid]).abs() - ref[mid].abs() * 2 ** -8).max()) < 2e-4
    ones = torch.zeros(M, K)
    ones[:, 0] = 1.0
    aux = xs[:, None].expand(M, N).contiguous()
    g = ops.linear(ones.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16), None, act=Nn.ACT_MUL_DGELU, aux_in=aux.to(dev, torch.bfloat16), tile=tile).float().cpu()
    a64 = xs.double()
    dg = (0.5 * (1 + torch.erf(a64 / math.sqrt(2))) + a64 * torch.exp(-0.5 * a64 * a64) / math.sqrt(2 * math.pi))[:, None]
    assert float(g.min()) > -0.14 and float(g.max()) < 1.14
    assert float(((g.double() - dg).abs() - dg.abs() * 2 ** -8).max()) < 1e-3
    assert float(g[lo].abs().max()) < 1e-4 and float((g[hi] - 1.0).abs().max()) == 0.0  # (x * phi(x) = -7e-5 at -4.5: exact, not noise)


# tile: 0 = the library's choice; 128128 / 256256 / 320256 = that kernel forced (256256 / 320256 are the persistent ping-pong kernel
# with 256- / 320-row tiles that every bench-size GEMM runs on; theia_gemm_nt refuses the request instead of falling back, so a
# passing forced case ran that kernel).  The f32 instantiation of the ping-pong kernel shares all of its indexing, zero-page,
# tile-walk and epilogue code with the bf16 one and is held to 1e-4; the 320-row tiles are bf16 only.
TILES = [0, 128128, 256256, 320256]


def _skip_f32_320(dt, tile):
    if tile == 320256 and dt != torch.bfloat16:
        pytest.skip("320-row tiles: bf16 instantiations only")


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(197 * 3, 192, 192), (300, 576, 192), (129, 32, 256), (1000, 1280, 384), (64, 768, 3072), (77, 256, 32),
                                   (2600, 768, 768), (513, 264, 96), (256 * 5, 512, 64)])
def test_linear_bias_epilogues(dt, M, N, K, tile):
    from theia_amd import ops, _native as Nn
    from functools import partial
    dev = _dev()
    _skip_f32_320(dt, tile)
    if tile == 128128 and Nn.lib().theia_gemm_nt_tile(M, N, Nn.dtype_code(dt)) == 128064:
        pytest.skip("narrow N: the 128x64 tile is the library's own choice")
    linear = partial(ops.linear, tile=tile)
    x = h((M, K), 1, 1.0)
    w = h((N, K), 2, 1.0 / math.sqrt(K))
    bias = h((N,), 3, 0.1)
    res = h((M, N), 4, 1.0)
    xr, wr, rr = rnd(x, dt), rnd(w, dt), rnd(res, dt)
    ref = xr @ wr.t() + bias
    xd, wd, bd, rd = x.to(dev, dt), w.to(dev, dt), bias.to(dev), res.to(dev, dt)
    bf = dt == torch.bfloat16  # bf16: every entry also within one bf16 rounding of ITS OWN magnitude (+ accumulation noise)
    y = linear(xd, wd, bd)
    assert relerr(y.float(), ref) < TOL[dt]
    assert not bf or elem_excess(y, ref) < ELEM_FLOOR
    y = linear(xd, wd, bd, resid=rd)
    assert relerr(y.float(), ref + rr) < TOL[dt]
    assert not bf or elem_excess(y, ref + rr) < ELEM_FLOOR
    pre = torch.empty(M, N, dtype=dt, device=dev)
    y = linear(xd, wd, bd, act=Nn.ACT_GELU, aux_out=pre)
    assert relerr(pre.float(), ref) < TOL[dt]
    assert relerr(y.float(), torch.nn.functional.gelu(ref)) < TOL[dt]
    assert not bf or max(elem_excess(pre, ref), elem_excess(y, torch.nn.functional.gelu(ref.double()))) < ELEM_FLOOR
    y = linear(xd, wd, bd, act=Nn.ACT_RELU)
    assert relerr(y.float(), torch.relu(ref)) < TOL[dt]
    assert not bf or elem_excess(y, torch.relu(ref)) < ELEM_FLOOR
    # backward-of-GELU epilogue
    aux = h((M, N), 5, 2.0)
    ar = rnd(aux, dt)
    y = linear(xd, wd, None, act=Nn.ACT_MUL_DGELU, aux_in=aux.to(dev, dt))
    a64 = ar.double()
    dg = 0.5 * (1 + torch.erf(a64 / math.sqrt(2))) + a64 * torch.exp(-0.5 * a64 * a64) / math.sqrt(2 * math.pi)
    assert relerr(y.float(), (xr @ wr.t()).double() * dg) < TOL[dt]
    # backward-of-ReLU epilogue
    y = linear(xd, wd, None, act=Nn.ACT_MUL_DRELU, aux_in=aux.to(dev, dt))
    assert relerr(y.float(), (xr @ wr.t()) * (ar > 0)) < TOL[dt]
    # strided output + residual read from the output buffer itself (how dz[:, 0] accumulates the CLS heads)
    wide = torch.zeros(M, N + 64, dtype=dt, device=dev)
    wide[:, 64:] = rd
    linear(xd, wd, bd, out=wide[:, 64:], resid=wide[:, 64:])
    assert relerr(wide[:, 64:].float(), ref + rr) < TOL[dt]
    assert float(wide[:, :64].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(197 * 4, 192, 192), (1000, 128, 256), (4096 + 37, 64, 64), (513, 1280, 192), (300, 32, 384), (2049, 24, 128), (777, 8, 64),
                                   (197 * 9, 768, 768), (2000, 3072, 768), (5000, 256, 768), (96, 1280, 768),
                                   (1500, 384, 384), (777, 1536, 384), (900, 384, 1536), (600, 576, 192), (1111, 192, 768), (650, 1024, 320),
                                   (262144 + 96, 32, 384)])  # skinny output over very many rows (the Depth head at b = 64: ping-pong kernel, 256 splits)
def test_linear_wgrad_and_colsum(dt, M, N, K):
    from theia_amd import ops
    dev = _dev()
    dy = h((M, N), 11, 1.0)
    x = h((M, K), 12, 1.0)
    dyr, xr = rnd(dy, dt), rnd(x, dt)
    ref = dyr.t().double() @ xr.double()
    g = torch.full((N, K), 0.5, dtype=torch.float32, device=dev)
    ops.linear_wgrad(dy.to(dev, dt), x.to(dev, dt), g, accumulate=True)
    assert relerr(g, ref + 0.5) < 1e-4 * (1 if dt == torch.float32 else 10)
    ops.linear_wgrad(dy.to(dev, dt), x.to(dev, dt), g, accumulate=False)
    assert relerr(g, ref) < 1e-4 * (1 if dt == torch.float32 else 10)
    cs = torch.zeros(N, dtype=torch.float32, device=dev)
    ops.colsum(dy.to(dev, dt), cs, accumulate=False)
    assert relerr(cs, dyr.double().sum(0)) < 1e-5
    # bias gradient through the weight-gradient entry point (fused into the ping-pong GEMM where it runs, colsum elsewhere),
    # with accumulation, on a column slice of a wider dY (the q/k/v slices of dQKV)
    wide = torch.zeros(M, N + 64, dtype=dt, device=dev)
    wide[:, 64:] = dy.to(dev, dt)
    gb = torch.full((N,), 0.25, dtype=torch.float32, device=dev)
    g2 = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ops.linear_wgrad(wide[:, 64:], x.to(dev, dt), g2, accumulate=False, bias=(gb, True))
    assert relerr(g2, ref) < 1e-4 * (1 if dt == torch.float32 else 10)
    assert relerr(gb - 0.25, dyr.double().sum(0)) < 1e-5
    ops.linear_wgrad(wide[:, 64:], x.to(dev, dt), g2, accumulate=False, bias=(gb, False))
    assert relerr(gb, dyr.double().sum(0)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,D", [(197 * 8, 768), (197 * 5 + 3, 384), (1000, 192), (700, 256)])
def test_grouped_linear_wgrad_matches_the_single_launches(dt, M, D):
    """theia_gemm_wgrad_group (ABI v11): the fused q/k/v gradient ([3D, D], a column slice of nothing -- dense) and o_proj's ([D, D]) in one
    launch with one split count, each with its bias gradient, accumulation on / off per problem; f32 (no grouped kernel) takes the
    one-by-one path through the same entry point.  Against f64, and against ops.linear_wgrad of each problem."""
    from theia_amd import ops
    dev = _dev()
    dq, a = h((M, 3 * D), 21, 1.0), h((M, D), 22, 1.0)
    dh, o = h((M, D), 23, 1.0), h((M, D), 24, 1.0)
    tol = 1e-4 if dt == torch.float32 else 1e-3
    probs, refs = [], []
    for dy, x, acc in ((dq, a, False), (dh, o, True)):
        dyr, xr = rnd(dy, dt), rnd(x, dt)
        gw = torch.full((dy.shape[1], x.shape[1]), 0.5, dtype=torch.float32, device=dev)
        gb = torch.full((dy.shape[1],), 0.25, dtype=torch.float32, device=dev)
        probs.append((dy.to(dev, dt), x.to(dev, dt), gw, acc, (gb, acc)))
        refs.append((dyr.t().double() @ xr.double() + (0.5 if acc else 0.0), dyr.double().sum(0) + (0.25 if acc else 0.0)))
    ops.linear_wgrad_group(probs)
    for (dy, x, gw, acc, (gb, _)), (rw, rb) in zip(probs, refs):
        assert relerr(gw, rw) < tol, (dy.shape, relerr(gw, rw))
        assert relerr(gb, rb) < 1e-5
        g1 = torch.full_like(gw, 0.5)
        b1 = torch.full_like(gb, 0.25)
        ops.linear_wgrad(dy, x, g1, acc, bias=(b1, acc))
        assert relerr(gw, g1) < (1e-6 if dt == torch.float32 else 2e-4)  # same products, another split of the f32 row sums
        assert relerr(gb, b1) < 1e-6


def test_grouped_wgrad_finish_is_bit_identical_to_the_single_reductions():
    """theia_wgrad_finish_group (ABI v12): the slab reductions behind one grouped weight-gradient launch as ONE launch -- the same sums in the
    same order as theia_wgrad_finish per problem (bitwise), accumulate on / off per job, bias partials included; a job outside the
    16-byte path is refused without launching anything; theia_transpose_acc2_f32 against two theia_transpose_acc_f32."""
    import ctypes
    from theia_amd import ops, _native as Nn
    dev = _dev()
    torch.manual_seed(5)
    splits = 7
    shapes = [(576, 192), (192, 192), (768, 192), (192, 768)]
    jobs = (Nn.WgradFinishJob * len(shapes))()
    keep, outs = [], []
    for i, (n, c) in enumerate(shapes):
        slabs, bsl = torch.randn(splits, n, c, device=dev), torch.randn(splits, n, device=dev)
        acc = bool(i & 1)
        gw, gb = torch.randn(n, c, device=dev), torch.randn(n, device=dev)
        w1, b1 = gw.clone(), gb.clone()
        ops.wgrad_finish(slabs, splits, n, 1, c, w1, c, 0, 1, acc, (bsl, b1, not acc))
        jobs[i] = Nn.WgradFinishJob(slabs.data_ptr(), gw.data_ptr(), bsl.data_ptr(), gb.data_ptr(), c, n, c, int(acc), int(not acc))
        keep.append((slabs, bsl))
        outs.append((gw, gb, w1, b1))
    assert Nn.lib().theia_wgrad_finish_group(jobs, len(shapes), splits, Nn.stream_ptr()) == 0
    for gw, gb, w1, b1 in outs:
        assert torch.equal(gw, w1) and torch.equal(gb, b1)
    # weights only (no bias pointers), 2 jobs
    j2 = (Nn.WgradFinishJob * 2)()
    o2 = []
    for i, (n, c) in enumerate(shapes[:2]):
        gw = torch.zeros(n, c, device=dev)
        w1 = torch.zeros(n, c, device=dev)
        ops.wgrad_finish(keep[i][0], splits, n, 1, c, w1, c, 0, 1, False)
        j2[i] = Nn.WgradFinishJob(keep[i][0].data_ptr(), gw.data_ptr(), None, None, c, n, c, 0, 0)
        o2.append((gw, w1))
    assert Nn.lib().theia_wgrad_finish_group(j2, 2, splits, Nn.stream_ptr()) == 0
    for gw, w1 in o2:
        assert torch.equal(gw, w1)
    # an output that is not 16-byte aligned: refused, nothing written
    base = torch.zeros(192 * 192 + 1, device=dev)
    j2[1] = Nn.WgradFinishJob(keep[1][0].data_ptr(), base[1:].data_ptr(), None, None, 192, 192, 192, 0, 0)
    assert Nn.lib().theia_wgrad_finish_group(j2, 2, splits, Nn.stream_ptr()) == Nn.ERR_UNSUPPORTED
    assert float(base.abs().sum()) == 0.0
    assert Nn.lib().theia_wgrad_finish_group(j2, 5, splits, Nn.stream_ptr()) == Nn.ERR_UNSUPPORTED
    # paired transpose
    R, Cc = 196, 192
    a, b_ = torch.randn(R, Cc, device=dev), torch.randn(R, Cc, device=dev)
    d0, d1 = torch.randn(Cc, R, device=dev), torch.randn(Cc, R, device=dev)
    e0, e1 = d0.clone(), d1.clone()
    ops.transpose_acc(a, e0, R, Cc, True)
    ops.transpose_acc(b_, e1, R, Cc, False)
    ops.transpose_acc2(a, d0, True, b_, d1, False, R, Cc)
    assert torch.equal(d0, e0) and torch.equal(d1, e1) and torch.equal(d1, b_.t())


def _nhwc(a):
    return torch.from_numpy(a).permute(0, 2, 3, 1).contiguous()


def _pack(plan_pack, w, dt, dev):
    from theia_amd import ops
    d0, d1, d2, s0, s1, s2 = plan_pack
    out = torch.empty(d0 * d1 * d2, dtype=dt, device=dev)
    ops.cast_permute3(w.to(dev), out, d0, d1, d2, s0, s1, s2)
    return out


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["conv_p1", "convT_s1", "convT_s2_p1", "convT_s2_op1"])
@pytest.mark.parametrize("C", [64, 256, 384, 192])  # the ping-pong WEIGHT-GRADIENT kernel: in_c >= 128; 384 / 192: a partial last c tile
@pytest.mark.parametrize("tile", TILES + [256009])  # forward / data-gradient GEMMs: library's choice, 2-stage 128x128, 256x256
# ping-pong, and the one-image-per-tile convolution kernel (256009: the row maps it takes -- 16x16 output, stride 1, 3x3)
def test_conv_family_fwd_dgrad_wgrad(dt, kind, C, tile):
    """Implicit-GEMM convolutions vs the oracle's shifted-matmul restatement (itself pinned to torch by G10)."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    _skip_f32_320(dt, tile)
    b = 3
    IH = {"conv_p1": 16, "convT_s1": 14, "convT_s2_p1": 16, "convT_s2_op1": 31}[kind]
    x = h((b, IH, IH, C), 21, 1.0)
    W = h((C, C, 3, 3), 22, 1.0 / math.sqrt(9 * C))
    bias = h((C,), 23, 0.1)
    xr, Wr = rnd(x, dt), rnd(W, dt)
    xr.requires_grad_(True)
    Wr.requires_grad_(True)
    if kind == "conv_p1":
        plan = ops.plan_conv3x3(C, IH)
        ref = O.conv3x3_p1(xr, Wr, bias)
    else:
        s, p, op = {"convT_s1": (1, 0, 0), "convT_s2_p1": (2, 1, 0), "convT_s2_op1": (2, 0, 1)}[kind]
        plan = ops.plan_convT3x3(C, IH, s, p, op)
        ref = O.convT3x3(xr, Wr, bias, s, p, op)
    OH = plan.out_hw
    assert ref.shape == (b, OH, OH, C)
    ref_relu = torch.relu(ref)
    gy = h((b, OH, OH, C), 24, 1.0)
    gyr = rnd(gy, dt)
    (ref * gyr).sum().backward()
    xd = x.to(dev, dt)
    wf = _pack(plan.pack_fwd, W, dt, dev)
    wdg = _pack(plan.pack_dgrad, W, dt, dev)
    out = torch.empty(b, OH, OH, C, dtype=dt, device=dev)
    out_relu = torch.empty_like(out)
    conv_kernel_fwd = kind in ("conv_p1", "convT_s1")  # 16x16 outputs of a stride-1 3x3 map
    conv_kernel_dgrad = kind == "conv_p1"
    for rmap, mpi in plan.fwd:
        assert (ops.gemm_nt(xd, wf, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, tile=256009, plan_only=True) == 256009) == conv_kernel_fwd
    if tile == 256009 and not conv_kernel_fwd:
        with pytest.raises(Nn.TheiaNativeError, match="256009"):
            rmap, mpi = plan.fwd[0]
            ops.gemm_nt(xd, wf, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, tile=tile)
        return
    for rmap, mpi in plan.fwd:
        ops.gemm_nt(xd, wf, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, bias=bias.to(dev), tile=tile)
        ops.gemm_nt(xd, wf, out_relu, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, bias=bias.to(dev), act=Nn.ACT_RELU, tile=tile)
    assert relerr(out.float(), ref.detach()) < TOL[dt]
    assert relerr(out_relu.float(), ref_relu.detach()) < TOL[dt]
    if dt == torch.bfloat16:  # ... and every entry within one bf16 rounding of its own magnitude (small entries included)
        assert max(elem_excess(out, ref.detach()), elem_excess(out_relu, ref_relu.detach())) < ELEM_FLOOR
    if tile == 256009 and not conv_kernel_dgrad:
        return
    # data gradient
    gyd = gy.to(dev, dt)
    dx = torch.empty(b, IH, IH, C, dtype=dt, device=dev)
    rmap, mpi = plan.dgrad
    ops.gemm_nt(gyd, wdg, dx, b * mpi, C, 9 * C, rmap, 9 * C, C, tile=tile)
    assert relerr(dx.float(), xr.grad) < TOL[dt]
    assert dt != torch.bfloat16 or elem_excess(dx, xr.grad) < ELEM_FLOOR
    if tile != 0:
        return  # the weight-gradient kernels have no tile request: covered once
    # weight gradient: reduction over output pixels (all classes into one slab set, then one reduce) ...
    Mtot = b * OH * OH
    splits = ops.wgrad_splits(Mtot, C, 9 * C)
    slabs = torch.empty(splits * C * 9 * C, dtype=torch.float32, device=dev)
    for rmap, mpi in plan.fwd:
        ops.gemm_wgrad(gyd, xd, slabs, b * mpi, C, C, 9, splits, rmap)
    gw = torch.zeros(C, C, 3, 3, dtype=torch.float32, device=dev)
    sn, ss, sc = plan.grad_strides
    ops.wgrad_reduce(slabs, splits, C, 9, C, gw, sn, ss, sc, accumulate=False)
    assert relerr(gw, Wr.grad) < TOL[dt]
    # ... and the engine's entry point (stride-2 transposed convs reduce over input pixels instead), with accumulation
    gw2 = torch.full((C, C, 3, 3), 0.5, dtype=torch.float32, device=dev)
    gb2 = torch.full((C,), 0.5, dtype=torch.float32, device=dev)
    ops.conv_wgrad(plan, gyd.view(b, -1), xd.view(b, -1), b, C, gw2, accumulate=True, bias=(gb2, True))
    assert plan.wgrad_swapped == kind.startswith("convT_s2")
    assert relerr(gw2 - 0.5, Wr.grad) < TOL[dt]
    assert relerr(gb2 - 0.5, gyr.double().sum((0, 1, 2))) < 1e-5


@pytest.mark.parametrize("kind", ["conv_p1", "convT_s1", "conv_p1_8x8"])
@pytest.mark.parametrize("b", [29, 37])
def test_conv_wgrad_periodic_rows_with_uneven_splits(b, kind):
    """The weight-gradient kernel's periodic row mode (images of a multiple of 32 pixels whose width divides 32: a step of 32 rows
    never wraps in x, the y wrap falls on the step that re-enters an image, and whether a tap's input pixel exists repeats with the
    image's steps; each staged row tests one bit of a mask built once).  conv_p1: dense 16x16 input (period 8); convT_s1: 14x14 input
    under a 16x16 output (non-zero image-wrap constant on the input side); conv_p1_8x8: 8x8 maps (period 2, four image rows per
    step).  The image counts give splits whose number of steps is not a multiple of the period, so the splits start at every phase
    (the 3-image case of test_conv_family_fwd_dgrad_wgrad only ever starts at phase 0), and the last split is ragged.  Against the
    oracle's shifted-matmul convolution, for the bf16 ping-pong kernel and -- same rounded inputs -- the exact-f32 2-stage kernel."""
    from theia_amd import ops
    dev = _dev()
    C, dt = 256, torch.bfloat16
    IH = {"conv_p1": 16, "convT_s1": 14, "conv_p1_8x8": 8}[kind]
    x = h((b, IH, IH, C), 31, 1.0)
    W = h((C, C, 3, 3), 32, 1.0 / math.sqrt(9 * C))
    xr, Wr = rnd(x, dt), rnd(W, dt)
    Wr.requires_grad_(True)
    if kind == "convT_s1":
        plan, ref = ops.plan_convT3x3(C, IH, 1, 0, 0), O.convT3x3(xr, Wr, torch.zeros(C), 1, 0, 0)
    else:
        plan, ref = ops.plan_conv3x3(C, IH), O.conv3x3_p1(xr, Wr, torch.zeros(C))
    OH = plan.out_hw
    assert ref.shape == (b, OH, OH, C) and not plan.wgrad_swapped and (OH * OH) % 32 == 0
    gy = h((b, OH, OH, C), 34, 1.0)
    (ref * rnd(gy, dt)).sum().backward()
    M = b * OH * OH
    splits = ops.wgrad_splits(M, C, 9 * C)
    steps_per_split, period = -(-(M // 32) // splits), OH * OH // 32
    assert splits > 1 and steps_per_split % period != 0  # uneven: the splits do not start at phase 0
    for t in (dt, torch.float32):
        gw = torch.zeros(C, C, 3, 3, dtype=torch.float32, device=dev)
        ops.conv_wgrad(plan, rnd(gy, dt).to(dev, t).view(b, -1), xr.detach().to(dev, t).view(b, -1), b, C, gw, accumulate=False)
        assert relerr(gw, Wr.grad) < (1e-4 if t == torch.float32 else 2e-3), t  # same bf16-rounded inputs, f32 accumulation in both


@pytest.mark.parametrize("tile", TILES + [256009])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pad_convT_on_strided_tokens(dt, tile):
    """The 14->16 pad reads z[:,1:,:] in place (batch stride 197*C, offset C) and its dgrad writes dz[:,1:,:]."""
    from theia_amd import ops
    dev = _dev()
    _skip_f32_320(dt, tile)
    C, b = 64, 2
    z = h((b, 197, C), 31, 1.0)
    W = h((C, C, 3, 3), 32, 0.05)
    bias = h((C,), 33, 0.1)
    zr, Wr = rnd(z, dt), rnd(W, dt)
    ref = O.convT3x3(zr[:, 1:].reshape(b, 14, 14, C), Wr, bias, 1, 0, 0)
    plan = ops.plan_convT3x3(C, 14, 1, 0, 0, in_bs=197 * C, in_off=C)
    wf = _pack(plan.pack_fwd, W, dt, dev)
    out = torch.empty(b, 16, 16, C, dtype=dt, device=dev)
    rmap, mpi = plan.fwd[0]
    ops.gemm_nt(z.to(dev, dt), wf, out, b * mpi, C, 9 * C, rmap, 9 * C, C, bias=bias.to(dev), tile=tile)
    assert relerr(out.float(), ref) < TOL[dt]
    gy = h((b, 16, 16, C), 34, 1.0)
    dz = torch.zeros(b, 197, C, dtype=dt, device=dev)
    wdg = _pack(plan.pack_dgrad, W, dt, dev)
    rmap, mpi = plan.dgrad
    if tile == 256009:  # the data-gradient has 14x14 rows per image: not a one-image-per-tile map
        assert ops.gemm_nt(gy.to(dev, dt), wdg, dz, b * mpi, C, 9 * C, rmap, 9 * C, C, plan_only=True) != 256009
        return
    ops.gemm_nt(gy.to(dev, dt), wdg, dz, b * mpi, C, 9 * C, rmap, 9 * C, C, tile=tile)
    zz = zr.clone().requires_grad_(True)
    (O.convT3x3(zz[:, 1:].reshape(b, 14, 14, C), Wr, bias, 1, 0, 0) * rnd(gy, dt)).sum().backward()
    assert relerr(dz.float(), zz.grad) < TOL[dt]
    assert float(dz[:, 0].float().abs().max()) == 0.0
    # the engine accumulates every head's data-gradient into the same dz: resid = out
    ops.gemm_nt(gy.to(dev, dt), wdg, dz, b * mpi, C, 9 * C, rmap, 9 * C, C, resid=dz, tile=tile)
    assert relerr(dz.float(), 2 * zz.grad) < TOL[dt]
    assert float(dz[:, 0].float().abs().max()) == 0.0


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [192, 768])
def test_patch_embedding_gemm_rowtab_and_token_rows(dt, D, tile):
    """Patch GEMM as the engine launches it (engine._backbone_fwd): plain [b*196, 768] patches -> rows 1 + p of the [b, 197, D]
    token matrix, bias + position-embedding row table (period 196) in the epilogue; row 0 of every image is left alone."""
    from theia_amd import ops
    dev = _dev()
    _skip_f32_320(dt, tile)
    b = 5
    patches = h((b * 196, 768), 35, 1.0)
    w = h((D, 768), 36, 1.0 / math.sqrt(768))
    bias = h((D,), 37, 0.1)
    pos = h((197, D), 38, 0.5)
    ref = (rnd(patches, dt) @ rnd(w, dt).t() + bias).view(b, 196, D) + pos[1:]
    rmap = ops.rowmap([(0, 0, 0)], (14, 14), (14, 14), 1, 768, 196 * 768, 0, 14, 1, 0, 0, 197 * D, D)
    out = torch.full((b, 197, D), 7.0, dtype=dt, device=dev)
    from theia_amd import _native as Nn
    if tile in (256256, 320256):  # the persistent ping-pong kernel carries no position row table: refused, not replaced
        with pytest.raises(Nn.TheiaNativeError, match="ping-pong"):
            ops.gemm_nt(patches.to(dev, dt), w.to(dev, dt), out, b * 196, D, 768, rmap, 768, D, bias=bias.to(dev),
                        rowtab=pos.to(dev)[1:], rowtab_period=196, tile=tile)
        return
    ops.gemm_nt(patches.to(dev, dt), w.to(dev, dt), out, b * 196, D, 768, rmap, 768, D, bias=bias.to(dev),
                rowtab=pos.to(dev)[1:], rowtab_period=196, tile=tile)
    assert relerr(out[:, 1:].float(), ref) < TOL[dt]
    assert float((out[:, 0].float() - 7.0).abs().max()) == 0.0


def test_pingpong_matches_2stage_up_to_rare_ulp_flips():
    """The persistent ping-pong kernel (256- and 320-row tiles, several tiles per workgroup at these sizes: B = 64 of the base model)
    against the 2-stage 128x128 kernel on the same bf16 operands: same products, f32 accumulation started from the bias / residual
    row instead of ending with it -> the bf16 outputs are identical except for single-ulp differences in a small fraction of the
    elements (< 1e-3; measured 1e-4 .. 5e-4).  Covers the single-tap and multi-tap instantiations, the statistics epilogue, the
    in-place residual of the pad data-gradient, GELU with its saved pre-activation."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    dt = torch.bfloat16
    torch.manual_seed(0)
    C, b = 768, 64

    def close(a, ref, what):
        a32, r32 = a.float(), ref.float()
        frac = float((a32 != r32).float().mean())
        # one bf16 ulp of the larger magnitude (2^-7 relative); values that cancel to ~0 differ by the f32 noise of the partial sums instead
        ulp = torch.maximum(a32.abs(), r32.abs()) * 2.0 ** -7 + 1e-5 * float(r32.abs().max())
        worst = float(((a32 - r32).abs() / ulp).max())
        assert frac < 1e-3 and worst <= 1.0 + 1e-6, (what, frac, worst)

    W = (torch.randn(C, 9 * C, device=dev) / math.sqrt(9 * C)).to(dt)
    bias = torch.randn(C, device=dev) * 0.1
    plan = ops.plan_convT3x3(C, 14, 1, 0, 0, in_bs=197 * C, in_off=C)  # the 14 -> 16 pad on the token matrix
    z = torch.randn(b, 197 * C, device=dev).to(dt)
    (rmap, mpi), = plan.fwd
    outs, sums = {}, {}
    for tile in (128128, 256256, 320256):
        outs[tile] = torch.zeros(b, 256 * C, dtype=dt, device=dev)
        sums[tile] = torch.zeros(b, 2, dtype=torch.int64, device=dev)
        ops.gemm_nt(z, W, outs[tile], b * mpi, C, 9 * C, rmap, 9 * C, C, bias=bias, tile=tile, ln_sums=sums[tile])
    for tile in (256256, 320256):
        close(outs[tile], outs[128128], ("pad fwd", tile))
        # sum of squares (the first moment of this output is ~0): same to 1e-5
        assert relerr(sums[tile][:, 1].double(), sums[128128][:, 1].double()) < 1e-5
    rmap, mpi = plan.dgrad
    gy = torch.randn(b, 256 * C, device=dev).to(dt)
    dz0 = torch.randn(b, 197 * C, device=dev).to(dt)
    res = {}
    for tile in (128128, 256256, 320256):
        res[tile] = dz0.clone()
        ops.gemm_nt(gy, W, res[tile], b * mpi, C, 9 * C, rmap, 9 * C, C, resid=res[tile], tile=tile)  # in place, as the engine does
    for tile in (256256, 320256):
        close(res[tile], res[128128], ("pad dgrad + residual", tile))
    for (M, N, K, kind) in ((b * 197, 3072, 768, "gelu"), (b * 197, 768, 3072, "resid"), (b * 4096, 256, 768, "plain")):
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
        bb_ = torch.randn(N, device=dev) * 0.1
        r = torch.randn(M, N, device=dev).to(dt)
        ys = {}
        for tile in (128128, 256256, 320256):
            if kind == "gelu":
                pre = torch.empty(M, N, dtype=dt, device=dev)
                ys[tile] = (ops.linear(x, w, bb_, act=Nn.ACT_GELU, aux_out=pre, tile=tile), pre)
            else:
                ys[tile] = (ops.linear(x, w, bb_, resid=r if kind == "resid" else None, tile=tile),)
        for tile in (256256, 320256):
            for i, y in enumerate(ys[tile]):
                close(y, ys[128128][i], (kind, i, tile))


def test_conv_kernel_column_split_at_bench_size():
    """The 3x3 convolution kernel at the bench's own size (128 images x 768 channels: 384 tiles of 256 columns = 1.5 rounds of the
    chip), where its launcher splits the columns into 256 full tiles + 256 tiles of 128 columns (gemm_conv_pp_kernel<.., 128>):
    forward with ReLU + LayerNorm statistics and the data-gradient with an in-place residual, against the 2-stage 128x128 kernel on
    the same operands (different tap order in the f32 sums: single-ulp differences in a small fraction of the elements)."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    dt = torch.bfloat16
    torch.manual_seed(3)
    C, b = 768, 128
    plan = ops.plan_conv3x3(C, 16)
    (fmap, mpi), = plan.fwd
    x = torch.randn(b, 256 * C, device=dev).to(dt)
    W = (torch.randn(C, 9 * C, device=dev) / math.sqrt(9 * C)).to(dt)
    bias = torch.randn(C, device=dev) * 0.1
    assert ops.gemm_nt(x, W, x, b * mpi, C, 9 * C, fmap, 9 * C, C, plan_only=True) == 256009

    def close(a, ref, what, frac_max):
        a32, r32 = a.float(), ref.float()
        frac = float((a32 != r32).float().mean())
        ulp = torch.maximum(a32.abs(), r32.abs()) * 2.0 ** -7 + 1e-5 * float(r32.abs().max())
        worst = float(((a32 - r32).abs() / ulp).max())
        assert frac < frac_max and worst <= 1.0 + 1e-6, (what, frac, worst)

    outs, sums = {}, {}
    for tile in (128128, 256009):
        outs[tile] = torch.zeros(b, 256 * C, dtype=dt, device=dev)
        sums[tile] = torch.zeros(b, 2, dtype=torch.int64, device=dev)
        ops.gemm_nt(x, W, outs[tile], b * mpi, C, 9 * C, fmap, 9 * C, C, bias=bias, act=Nn.ACT_RELU, tile=tile, ln_sums=sums[tile])
    close(outs[256009], outs[128128], "conv fwd", 2e-3)
    assert relerr(sums[256009].double(), sums[128128].double()) < 1e-5
    dmap, mpi = plan.dgrad
    gy = torch.randn(b, 256 * C, device=dev).to(dt)
    acc0 = torch.randn(b, 256 * C, device=dev).to(dt)
    res = {}
    for tile in (128128, 256009):
        res[tile] = acc0.clone()
        ops.gemm_nt(gy, W, res[tile], b * mpi, C, 9 * C, dmap, 9 * C, C, resid=res[tile], tile=tile)
    close(res[256009], res[128128], "conv dgrad + residual", 2e-3)


def test_forced_tile_is_refused_not_replaced():
    """A 256256 request the ping-pong kernel cannot take is an error (never a silent fall-back to another kernel)."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    x = torch.zeros(64, 40, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(64, 40, dtype=torch.bfloat16, device=dev)
    with pytest.raises(Nn.TheiaNativeError, match="ping-pong"):
        ops.linear(x, w, tile=256256)
    with pytest.raises(Nn.TheiaNativeError, match="bad tile"):
        ops.linear(x, w, tile=64064)
    assert float(ops.linear(x, w, tile=0).float().abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,kind", [(25216, 768, 768, "resid"), (25216, 3072, 768, "gelu"), (25216, 768, 3072, "dgelu"),
                                         (32768, 1280, 768, "plain")])
def test_bench_size_linears_run_the_pingpong_kernel(M, N, K, kind):
    """The exact ViT GEMM shapes of the default bench (per-GPU batch 128: M = 128*197 = 98.5 tiles of 256 rows) with the
    library's own dispatch, bf16, against an f32 evaluation of the same bf16-rounded inputs."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    dt = torch.bfloat16
    probe = torch.empty(8, dtype=dt, device=dev)  # (the planner only looks at shapes and which pointers are set)
    assert ops.gemm_nt(probe, probe, probe, M, N, K, ops.rm_plain(K, K, N), K, N, plan_only=True) == (320256 if N == 768 else 256256)
    x = h((M, K), 1, 1.0)
    w = h((N, K), 2, 1.0 / math.sqrt(K))
    bias = h((N,), 3, 0.1)
    xr, wr = rnd(x, dt), rnd(w, dt)
    ref = xr @ wr.t() + bias
    xd, wd, bd = x.to(dev, dt), w.to(dev, dt), bias.to(dev)
    if kind == "resid":
        res = h((M, N), 4, 1.0)
        y = ops.linear(xd, wd, bd, resid=res.to(dev, dt))
        assert relerr(y.float(), ref + rnd(res, dt)) < TOL[dt]
    elif kind == "gelu":
        pre = torch.empty(M, N, dtype=dt, device=dev)
        y = ops.linear(xd, wd, bd, act=Nn.ACT_GELU, aux_out=pre)
        assert relerr(pre.float(), ref) < TOL[dt]
        assert relerr(y.float(), torch.nn.functional.gelu(ref)) < TOL[dt]
    elif kind == "dgelu":
        aux = rnd(h((M, N), 5, 2.0), dt)
        y = ops.linear(xd, wd, None, act=Nn.ACT_MUL_DGELU, aux_in=aux.to(dev, dt))
        dg = 0.5 * (1 + torch.erf(aux / math.sqrt(2))) + aux * torch.exp(-0.5 * aux * aux) / math.sqrt(2 * math.pi)
        assert relerr(y.float(), (xr @ wr.t()) * dg) < TOL[dt]
    else:
        assert relerr(ops.linear(xd, wd, bd).float(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [192, 384, 768])
@pytest.mark.parametrize("M", [197 * 3 + 1, 197 * 3])  # odd M: the two-rows-per-wave kernels (D <= 256) have a last row without a partner
def test_layernorm_rows(dt, D, M):
    from theia_amd import ops
    dev = _dev()
    x = h((M, D), 41, 2.0) + 0.3
    g = h((D,), 42, 0.2) + 1.0
    bta = h((D,), 43, 0.2)
    xr = rnd(x, dt).requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = bta.clone().requires_grad_(True)
    ref = O._layernorm_rows(xr, gr, br, 1e-12)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev, dt), g.to(dev), bta.to(dev), 1e-12)
    assert relerr(y.float(), ref.detach()) < (1e-5 if dt == torch.float32 else 1e-2)
    dy = h((M, D), 44, 1.0)
    dres = h((M, D), 45, 1.0)
    (ref * rnd(dy, dt)).sum().backward()
    dg = torch.zeros(D, device=dev)
    db = torch.zeros(D, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev, dt), x.to(dev, dt), g.to(dev), mean, rstd, dres.to(dev, dt), dg, db, accumulate=False)
    assert relerr(dx.float(), xr.grad + rnd(dres, dt)) < (1e-4 if dt == torch.float32 else 2e-2)
    assert relerr(dg, gr.grad) < 1e-4
    assert relerr(db, br.grad) < 1e-4
    x64 = rnd(x, dt).double()
    assert relerr(mean, x64.mean(1)) < 1e-5 and relerr(rstd, 1.0 / torch.sqrt(x64.var(1, unbiased=False) + 1e-12)) < 1e-4
    # without a residual-stream gradient, accumulating into the affine gradients
    dx0 = ops.layernorm_bwd(dy.to(dev, dt), x.to(dev, dt), g.to(dev), mean, rstd, None, dg, db, accumulate=True)
    assert relerr(dx0.float(), xr.grad) < (1e-4 if dt == torch.float32 else 2e-2)
    assert relerr(dg, 2 * gr.grad) < 1e-4 and relerr(db, 2 * br.grad) < 1e-4


@pytest.mark.parametrize("D", [192, 384, 768])
def test_fused_e4m3_side_outputs_equal_the_quantised_bf16_outputs(D):
    """The *_q8 entry points (ABI v11): a pass's e4m3 copy of its main output must be, byte for byte, theia_quantize_fp8 of the bf16 output it
    stores beside it (same scale), the bf16 output itself must not change, and a given amax slot ends at max(amax, max |output|).
    Row LayerNorm forward / backward (one- and two-rows-per-wave kernels), LayerNorm[C,H,W] one-pass forward and backward, loss gradient."""
    from theia_amd import ops
    dev = _dev()
    dt = torch.bfloat16
    M = 197 * 3
    sc = torch.tensor([3.0], device=dev)

    def check(y, y8, amax, plain):
        # (the side-output forms are separate instantiations: the compiler may contract their f32 expressions differently -- a bf16 ulp)
        assert relerr(y.float(), plain.float()) < 1e-2 and float((y != plain).float().mean()) < 0.01
        want = ops.quantize_fp8(y.view(-1, y.shape[-1]), sc)
        assert torch.equal(y8.view(torch.uint8).view(-1), want.view(torch.uint8).view(-1))
        if amax is not None:
            assert float(amax) == max(0.125, float(y.float().abs().max()))

    x, g, bta = h((M, D), 41, 2.0).to(dev, dt), (h((D,), 42, 0.2) + 1.0).to(dev), h((D,), 43, 0.2).to(dev)
    for with_amax in (True, False):
        am = torch.tensor([0.125], device=dev) if with_amax else None
        y8 = torch.empty(M, D, dtype=torch.float8_e4m3fn, device=dev)
        y, mean, rstd = ops.layernorm_fwd(x, g, bta, 1e-12, q8=(y8, sc, am))
        check(y, y8, am, ops.layernorm_fwd(x, g, bta, 1e-12)[0])
        dy, dres = h((M, D), 44, 1.0).to(dev, dt), h((M, D), 45, 1.0).to(dev, dt)
        for res in (dres, None):
            am = torch.tensor([0.125], device=dev) if with_amax else None
            dg, db, dg0, db0 = (torch.zeros(D, device=dev) for _ in range(4))
            d8 = torch.empty(M, D, dtype=torch.float8_e4m3fn, device=dev)
            dx = ops.layernorm_bwd(dy, x, g, mean, rstd, res, dg, db, accumulate=False, q8=(d8, sc, am))
            check(dx, d8, am, ops.layernorm_bwd(dy, x, g, mean, rstd, res, dg0, db0, accumulate=False))
            assert relerr(dg, dg0) < 1e-5 and relerr(db, db0) < 1e-5
    # LayerNorm[C,H,W]: the one-pass forward (statistics from the producer's fixed-point sums) and the backward
    b, H, C = 5, 16, D // 3
    E = H * H * C
    xc = torch.relu(h((b, E), 51, 2.0) + 0.2).to(dev, dt)
    gc, sc_ = (h((E,), 52, 0.2) + 1.0).to(dev), h((E,), 53, 0.2).to(dev)
    x64 = xc.double()
    sums = torch.stack([(x64.sum(1) * 2 ** 24).round(), ((x64 * x64).sum(1) * 2 ** 24).round()], 1).to(torch.int64).contiguous()
    am = torch.tensor([0.125], device=dev)
    y8 = torch.empty(b, E, dtype=torch.float8_e4m3fn, device=dev)
    y, stats = ops.layernorm_chw_fwd(xc, gc, sc_, 1e-5, sums=sums, q8=(y8, sc, am))
    check(y, y8, am, ops.layernorm_chw_fwd(xc, gc, sc_, 1e-5, sums=sums)[0])
    dyc = h((b, E), 54, 1.0).to(dev, dt)
    for mask in (False, True):
        am = torch.tensor([0.125], device=dev)
        dg, ds, dg0, ds0 = (torch.zeros(E, device=dev) for _ in range(4))
        d8 = torch.empty(b, E, dtype=torch.float8_e4m3fn, device=dev)
        dx = ops.layernorm_chw_bwd(dyc, xc, gc, stats, dg, ds, relu_mask=mask, accumulate=False, q8=(d8, sc, am))
        check(dx, d8, am, ops.layernorm_chw_bwd(dyc, xc, gc, stats, dg0, ds0, relu_mask=mask, accumulate=False))
        assert relerr(dg, dg0) < 1e-5 and relerr(ds, ds0) < 1e-5
    # loss gradient
    pred, tgt = h((b, E), 61, 1.0).to(dev, dt), h((b, E), 62, 1.0).to(dev, dt)
    _, coef = ops.distill_loss_fwd(pred, tgt)
    w = torch.tensor([0.2, 0.9, 0.1], device=dev)
    sc.fill_(2.0 ** 12)  # (loss gradients are ~1 / (b E))
    am = torch.tensor([0.0], device=dev)
    d8 = torch.empty(b, E, dtype=torch.float8_e4m3fn, device=dev)
    dp = ops.distill_loss_bwd(pred, tgt, coef, w, q8=(d8, sc, am))
    plain = ops.distill_loss_bwd(pred, tgt, coef, w)
    assert relerr(dp.float(), plain.float()) < 1e-2 and float(am) == float(dp.float().abs().max())
    assert torch.equal(d8.view(torch.uint8), ops.quantize_fp8(dp, sc).view(torch.uint8))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H", [(64, 16), (192, 31), (64, 64)])
def test_layernorm_chw(dt, C, H):
    from theia_amd import ops
    dev = _dev()
    b = 5
    x = torch.relu(h((b, H, H, C), 51, 2.0) + 0.2)
    g = h((C, H, H), 52, 0.2) + 1.0
    s = h((C, H, H), 53, 0.2)
    xr = rnd(x, dt).requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    sr = s.clone().requires_grad_(True)
    ref = O.layernorm_chw(xr, gr, sr)
    E = H * H * C
    g_nhwc = g.permute(1, 2, 0).contiguous().view(E).to(dev)
    s_nhwc = s.permute(1, 2, 0).contiguous().view(E).to(dev)
    xd = x.to(dev, dt).view(b, E)
    y, stats = ops.layernorm_chw_fwd(xd, g_nhwc, s_nhwc, 1e-5)
    assert relerr(y.float().view(b, H, H, C), ref.detach()) < (1e-4 if dt == torch.float32 else 1e-2)
    dy = h((b, H, H, C), 54, 1.0)
    (ref * rnd(dy, dt)).sum().backward()
    dg = torch.zeros(E, device=dev)
    ds = torch.zeros(E, device=dev)
    dx = ops.layernorm_chw_bwd(dy.to(dev, dt).view(b, E), xd, g_nhwc, stats, dg, ds, relu_mask=False, accumulate=False)
    assert relerr(dx.float().view(b, H, H, C), xr.grad) < (1e-4 if dt == torch.float32 else 2e-2)
    assert relerr(dg.view(H, H, C).permute(2, 0, 1), gr.grad) < 1e-4
    assert relerr(ds.view(H, H, C).permute(2, 0, 1), sr.grad) < 1e-4
    # relu mask folds d relu: x == relu(pre) so (x > 0) selects the live units
    dxm = ops.layernorm_chw_bwd(dy.to(dev, dt).view(b, E), xd, g_nhwc, stats, dg, ds, relu_mask=True, accumulate=False)
    assert relerr(dxm.float().view(b, H, H, C), xr.grad * (rnd(x, dt) > 0)) < (1e-4 if dt == torch.float32 else 2e-2)
    # the same pass also hands out the column sums of dx (the producing convolution's bias gradient): exactly the sums of the dx it
    # stores (f32 accumulation of the stored values), with and without accumulation
    cs = torch.full((C,), 0.25, dtype=torch.float32, device=dev)
    dx2 = ops.layernorm_chw_bwd(dy.to(dev, dt).view(b, E), xd, g_nhwc, stats, dg, ds, relu_mask=True, accumulate=False, dxsum=(cs, True))
    assert torch.equal(dx2, dxm)
    want = dxm.float().view(-1, C).double().sum(0)
    assert relerr(cs - 0.25, want) < 1e-5
    ops.layernorm_chw_bwd(dy.to(dev, dt).view(b, E), xd, g_nhwc, stats, dg, ds, relu_mask=True, accumulate=False, dxsum=(cs, False))
    assert relerr(cs, want) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hh", [3, 12])
def test_attention_fwd_bwd(dt, hh):
    from theia_amd import ops
    dev = _dev()
    b, n = 2, 197
    D = hh * 64
    qkv = h((b, n, 3 * D), 61, 1.5)
    # spike one key against one query so softmax has a dominant entry
    qkv[0, 5, :64] *= 4.0
    qkv[0, 9, D:D + 64] = qkv[0, 5, :64]
    r = rnd(qkv, dt).requires_grad_(True)
    q, k, v = r.split(D, dim=-1)
    q = q.view(b, n, hh, 64).transpose(1, 2)
    k = k.view(b, n, hh, 64).transpose(1, 2)
    v = v.view(b, n, hh, 64).transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) / 8.0, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(b, n, D)
    qd = qkv.to(dev, dt).view(b * n, 3 * D)
    o, lse = ops.attention_fwd(qd, b, n, hh)
    assert relerr(o.float().view(b, n, D), ref.detach()) < (1e-5 if dt == torch.float32 else 1e-2)
    do = h((b, n, D), 62, 1.0)
    (ref * rnd(do, dt)).sum().backward()
    dqkv = ops.attention_bwd(qd, o, do.to(dev, dt).view(b * n, D), lse, b, n, hh)
    assert relerr(dqkv.float().view(b, n, 3 * D), r.grad) < (1e-4 if dt == torch.float32 else 2e-2)


def test_attention_bwd_two_kernel_form_in_a_fresh_process():
    """The default bf16 backward is the single kernel (attn_bwd_fused_kernel); the two-kernel form stays selectable with
    THEIA_ATTN_BWD=split (read once per process): the same attention tests in a child process with that switch."""
    import subprocess
    import sys
    env = dict(os.environ, THEIA_ATTN_BWD="split")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_ops_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "test_attention_fwd_bwd or per_tensor_bound"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("sharp", [1.0, 3.0])
def test_attention_bwd_per_tensor_bound_at_batch_8(sharp):
    """The per-kernel bound behind the model-level bf16 gradient gates: the attention kernels alone, bf16, b = 8 x 12 heads x 197
    tokens, with the logits as they come (sharp = 1) and three times sharper (a near-one-hot softmax, P stored as bf16 for the PV /
    dV products), against an f64 evaluation of the same bf16 inputs.  Per tensor (dQ, dK, dV separately -- a max-norm over the whole
    dQKV matrix is blind to the small ones): cosine > 0.99999, norm-relative error < 5e-3 (measured: 0.999996-0.999997, 2.3e-3-2.7e-3).  The model-level cosines of 0.991-0.994
    (q_proj of the upper layers, the head LayerNorm affines) are NOT this kernel: they are the bf16 noise floor of a 12-layer
    network on near-zero gradients (test_model_gpu.py::test_bf16_bench_dispatch_agrees_with_the_2stage_kernels)."""
    from theia_amd import ops
    dev = _dev()
    dt = torch.bfloat16
    b, n, hh = 8, 197, 12
    D = hh * 64
    qkv = h((b, n, 3 * D), 161, 1.5)
    qkv[..., :2 * D] *= math.sqrt(sharp)  # q.k scales by `sharp`
    r = rnd(qkv, dt).double().requires_grad_(True)
    q, k, v = r.split(D, dim=-1)
    q = q.view(b, n, hh, 64).transpose(1, 2)
    k = k.view(b, n, hh, 64).transpose(1, 2)
    v = v.view(b, n, hh, 64).transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) / 8.0, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(b, n, D)
    do = rnd(h((b, n, D), 162, 1.0), dt)
    (ref * do.double()).sum().backward()
    qd = qkv.to(dev, dt).view(b * n, 3 * D)
    o, lse = ops.attention_fwd(qd, b, n, hh)
    a, bb_ = o.double().cpu().reshape(-1), ref.detach().reshape(-1)
    assert float((a - bb_).norm() / bb_.norm()) < 1e-2
    dqkv = ops.attention_bwd(qd, o, do.to(dev, dt).view(b * n, D), lse, b, n, hh).double().cpu().view(b, n, 3 * D)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, bb_ = dqkv[..., sl].reshape(-1), r.grad[..., sl].reshape(-1)
        cos = float((a @ bb_) / (a.norm() * bb_.norm()))
        err = float((a - bb_).norm() / bb_.norm())
        print(f"[attention bwd b=8 sharp={sharp}] {name}: cosine {cos:.6f}, |a-b|/|b| {err:.4f}, max softmax weight {float(p.max()):.3f}")
        assert cos > 0.99999 and err < 5e-3, (name, cos, err)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_distill_loss_against_oracle_and_golden(dt, golden_dir):
    from theia_amd import ops
    dev = _dev()
    g10 = np.load(os.path.join(golden_dir, "g10_micro_ops.npz"))
    b, E = 6, 256 * 40
    p = h((b, E), 71, 2.0)
    q = h((b, E), 72, 2.0)
    pr = rnd(p, dt).requires_grad_(True)
    losses = O.get_loss({"t": pr.view(b, 256, 40)}, {"t": q.view(b, 256, 40)})
    l, coef = ops.distill_loss_fwd(p.to(dev, dt), q.to(dev))
    got = l.cpu()
    assert abs(float(got[0]) - float(losses["mse_loss"])) < 1e-5 * abs(float(losses["mse_loss"]))
    assert abs(float(got[1]) - float(losses["cos_loss"])) < 1e-5
    assert abs(float(got[2]) - float(losses["l1_loss"])) < 1e-5 * abs(float(losses["l1_loss"]))
    for w in ([0.0, 0.9, 0.1], [1.0, 0.0, 0.0], [0.3, 0.5, 0.2]):
        pr.grad = None
        (w[0] * losses["mse_loss"] + w[1] * losses["cos_loss"] + w[2] * losses["l1_loss"]).backward(retain_graph=True)
        dp = ops.distill_loss_bwd(p.to(dev, dt), q.to(dev), coef, torch.tensor(w, device=dev))
        assert relerr(dp.float(), pr.grad) < (1e-4 if dt == torch.float32 else 1e-2)
    if dt == torch.float32:  # reference-produced micro golden (torch.nn losses as models/rvfm.py calls them)
        lp, lq = torch.from_numpy(g10["lp"]), torch.from_numpy(g10["lq"])
        # E must be a multiple of 8: tile each sample 4x (means / cosines are invariant to that)
        pp = lp.reshape(3, -1).repeat(1, 4).contiguous()
        qq = lq.reshape(3, -1).repeat(1, 4).contiguous()
        l, _ = ops.distill_loss_fwd(pp.to(dev), qq.to(dev))
        got = l.cpu()
        assert abs(float(got[0]) - float(g10["mse"])) < 1e-5 * float(g10["mse"])
        assert abs(float(got[1]) - float(g10["cos"])) < 1e-5
        assert abs(float(got[2]) - float(g10["smooth_l1"])) < 1e-5 * float(g10["smooth_l1"])


def test_distill_loss_with_bf16_teacher_features_is_bit_identical():
    """The reference's loader normalises the stored bf16 features in bf16 and widens them (data_utils.py:374-379): a bf16 target tensor
    holds the same values as the f32 one the reference feeds, so the bf16-target kernels (ABI v11) must give the same bits -- and
    anything but f32, or bf16 beside bf16 predictions, is refused."""
    from theia_amd import ops
    dev = _dev()
    b, E = 5, 4096 * 32 + 64
    p = h((b, E), 81, 2.0).to(dev, torch.bfloat16)
    q16 = h((b, E), 82, 2.0).to(dev, torch.bfloat16)
    q32 = q16.float()
    l32, c32 = ops.distill_loss_fwd(p, q32)
    l16, c16 = ops.distill_loss_fwd(p, q16)
    assert torch.equal(l32, l16) and torch.equal(c32, c16)
    w = torch.tensor([0.2, 0.9, 0.1], device=dev)
    assert torch.equal(ops.distill_loss_bwd(p, q32, c32, w), ops.distill_loss_bwd(p, q16, c16, w))
    with pytest.raises(RuntimeError):
        ops.distill_loss_fwd(p.float(), q16)  # bf16 targets beside f32 predictions


@pytest.mark.parametrize("channels_last", [True, False])
def test_patchify_bit_exact_indexing(channels_last):
    from theia_amd import ops
    dev = _dev()
    b = 3
    img = O.synth_images(b, seed=5)
    lut = torch.from_numpy(O.preprocess_lut())
    ref = O.patch_matrix(O.preprocess(img)).reshape(b * 196, 768)
    src = img if channels_last else img.permute(0, 3, 1, 2).contiguous()
    out = torch.empty(b * 196, 768, dtype=torch.float32, device=dev)
    ops.patchify(src.to(dev), lut.to(dev), out, channels_last)
    assert torch.equal(out.cpu(), ref)  # bit exact (table look-up + integer indexing)


def test_token_select_modes_and_feature_norm(golden_dir):
    from theia_amd import ops
    dev = _dev()
    b, n, D = 3, 197, 192
    x = h((b, n, D), 81, 1.0)
    xd = x.to(dev)
    for mode, name in ((0, None), (1, "mean_pooling"), (2, "max_pooling"), (3, "cls")):
        ref = O.handle_feature_output(x, name, 0)
        got = ops.token_select(xd, b, n, D, 0, mode).cpu()
        if mode == 1:
            assert relerr(got, ref) < 1e-6
        else:
            assert torch.equal(got, ref)
    ref = O.handle_feature_output(x, None, 3)
    assert torch.equal(ops.token_select(xd, b, n, D, 3, 0).cpu(), ref)
    g8 = np.load(os.path.join(golden_dir, "g8_feature_norm_bf16.npz"))
    xb = torch.from_numpy(g8["x_bits"]).view(torch.bfloat16)
    y = ops.feature_norm_bf16(xb.to(dev), torch.from_numpy(g8["mean"]).to(dev), torch.from_numpy(g8["std"]).to(dev)).cpu().numpy()
    assert np.array_equal(y, g8["y"])  # two bf16 roundings reproduced bit-exactly


@pytest.mark.parametrize("resample", [2, 3])
@pytest.mark.parametrize("hh,ww", [(50, 37), (300, 200), (224, 301), (301, 224), (448, 448), (17, 400), (1000, 750)])
def test_resize_u8_bit_exact(hh, ww, resample):
    """theia_resize_u8 == Pillow's Image.resize as the reference's processor calls it (oracle/pil_resize.py, pinned against
    Pillow and golden G12), bit for bit, both input layouts, both passes / single passes."""
    from oracle import pil_resize as R
    from theia_amd import ops
    dev = _dev()
    rng = np.random.default_rng(hh * 1000 + ww)
    img = rng.integers(0, 256, (2, hh, ww, 3), dtype=np.uint8)
    ref = np.stack([R.resize_u8(im, 224, 224, resample) for im in img])
    t = torch.from_numpy(img)
    got = ops.resize_u8(t.to(dev), True, 224, 224, resample)
    assert got.shape == (2, 224, 224, 3) and np.array_equal(got.cpu().numpy(), ref)
    got = ops.resize_u8(t.permute(0, 3, 1, 2).contiguous().to(dev), False, 224, 224, resample)
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("C,H", [(1280, 16), (1024, 16), (256, 64), (32, 64), (40, 7)])
def test_feature_ingest_bit_exact(C, H):
    """theia_feature_ingest_bf16 == decode_sample's rearrange + bf16 normalize_feature + .float() of the reference (oracle,
    itself pinned by golden G8), bit for bit; also through FeatureIngest's pinned staging + safetensors decode."""
    from safetensors.torch import save as sft_save
    from theia_amd import ops
    from theia_amd.dataset import FeatureIngest, decode_feature
    dev = _dev()
    b = 3
    x = (h((b, C, H, H), 51) * 3.0).to(torch.bfloat16)
    mean = h((C,), 52) * 0.5
    std = h((C,), 53).abs() + 0.5
    ref = O.ingest_feature_chw_bf16(x, mean, std)
    got = ops.feature_ingest_bf16(x.to(dev), mean.to(dev), std.to(dev))
    assert got.shape == (b, H * H, C) and torch.equal(got.cpu(), ref)
    assert torch.equal(ops.feature_ingest_bf16(x.to(dev), None, None).cpu(), O.ingest_feature_chw_bf16(x))
    ing = FeatureIngest(dev, {"t": mean}, {"t": std})
    samples = [decode_feature(sft_save({"embedding": x[i]}))["embedding"] for i in range(b)]
    for _ in range(2):  # second call reuses the pinned buffer
        out = ing({"t": samples})["t"]
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cast_batch_matches_single_casts(dt):
    """theia_cast_batch (one launch for the whole operand table) == the individual cast / transpose / permute kernels,
    bit for bit, including ragged shapes, strided destinations and f32 destinations."""
    from theia_amd import ops
    dev = _dev()
    C = 48
    w = h((100, 72), 41).to(dev)
    wq = h((3, 40, 40), 42).to(dev)
    cw = h((C, C, 3, 3), 43).to(dev)
    g = h((C, 7, 7), 44).to(dev)
    bias = h((77,), 45).to(dev)
    cb = ops.CastBatch(dev, dt)
    got, ref = {}, {}
    got["w"], ref["w"] = torch.empty(100, 72, dtype=dt, device=dev), torch.empty(100, 72, dtype=dt, device=dev)
    cb.add_cast(w, got["w"])
    ops.cast(w, ref["w"])
    got["wT"], ref["wT"] = torch.zeros(72, 100, dtype=dt, device=dev), torch.zeros(72, 100, dtype=dt, device=dev)
    cb.add_transpose(w, got["wT"])
    ops.cast_transpose(w, ref["wT"])
    got["qT"], ref["qT"] = torch.zeros(40, 120, dtype=dt, device=dev), torch.zeros(40, 120, dtype=dt, device=dev)
    for j in range(3):
        cb.add_transpose(wq[j], got["qT"][:, j * 40:], ldd=120)
        ops.cast_transpose(wq[j], ref["qT"][:, j * 40:], ldd=120)
    for name, pack in (("cf", ops.plan_conv3x3(C, 16).pack_fwd), ("cd", ops.plan_conv3x3(C, 16).pack_dgrad),
                       ("tf", ops.plan_convT3x3(C, 16, 2, 1, 0).pack_fwd), ("td", ops.plan_convT3x3(C, 16, 2, 1, 0).pack_dgrad)):
        got[name], ref[name] = torch.empty(C, 9 * C, dtype=dt, device=dev), torch.empty(C, 9 * C, dtype=dt, device=dev)
        cb.add(cw, got[name], *pack)
        ops.cast_permute3(cw, ref[name], *pack)
    got["g"], ref["g"] = torch.empty(49 * C, dtype=torch.float32, device=dev), torch.empty(49 * C, dtype=torch.float32, device=dev)
    cb.add(g, got["g"], 1, 49, C, 0, 1, 49)
    ops.cast_permute3(g, ref["g"], 49, 1, C, 1, 0, 49)
    got["b"], ref["b"] = torch.empty(77, dtype=torch.float32, device=dev), torch.empty(77, dtype=torch.float32, device=dev)
    cb.add_cast(bias, got["b"])
    ops.cast(bias, ref["b"])
    # ragged / unaligned shapes (scalar paths), a j-fast transpose whose last float4 is partial, and the real operand sizes
    wr = h((37, 50), 46).to(dev)
    got["rT"], ref["rT"] = torch.zeros(50, 37, dtype=dt, device=dev), torch.zeros(50, 37, dtype=dt, device=dev)
    cb.add_transpose(wr, got["rT"])
    ops.cast_transpose(wr, ref["rT"])
    wp = h((64, 70), 47).to(dev)
    got["pT"], ref["pT"] = torch.zeros(70, 64, dtype=dt, device=dev), torch.zeros(70, 64, dtype=dt, device=dev)
    cb.add_transpose(wp, got["pT"])
    ops.cast_transpose(wp, ref["pT"])
    big = h((768, 3072), 48).to(dev)
    got["bT"], ref["bT"] = torch.empty(3072, 768, dtype=dt, device=dev), torch.empty(3072, 768, dtype=dt, device=dev)
    cb.add_transpose(big, got["bT"])
    ops.cast_transpose(big, ref["bT"])
    cw2 = h((192, 192, 3, 3), 49).to(dev)
    for name, pack in (("cf2", ops.plan_conv3x3(192, 16).pack_fwd), ("cd2", ops.plan_conv3x3(192, 16).pack_dgrad)):
        got[name], ref[name] = torch.empty(192, 9 * 192, dtype=dt, device=dev), torch.empty(192, 9 * 192, dtype=dt, device=dev)
        cb.add(cw2, got[name], *pack)
        ops.cast_permute3(cw2, ref[name], *pack)
    g2 = h((64, 31, 31), 50).to(dev)  # [C, H, W] -> [HW, C] with HW = 961 (odd source pitch)
    got["g2"], ref["g2"] = torch.empty(961 * 64, dtype=torch.float32, device=dev), torch.empty(961 * 64, dtype=torch.float32, device=dev)
    cb.add(g2, got["g2"], 1, 961, 64, 0, 1, 961)
    ops.cast_permute3(g2, ref["g2"], 961, 1, 64, 1, 0, 961)
    cb.run()
    for k in got:
        assert torch.equal(got[k], ref[k]), k
    w.mul_(2.0)  # the table is reusable: same pointers, new values
    cb.run()
    ops.cast(w, ref["w"])
    assert torch.equal(got["w"], ref["w"])


def test_adamw_matches_torch():
    from theia_amd import ops
    dev = _dev()
    n = 10007
    p0 = h((n,), 91, 1.0)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = p0.to(dev).clone()
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        g = h((n,), 92 + step, 1.0)
        ref_p.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g.to(dev), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, step)
    assert relerr(p, ref_p.detach()) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["conv_p1", "convT_s1", "convT_s2_p1", "convT_s2_op1"])
@pytest.mark.parametrize("tile", [0, 128128, 256256, 320256])
def test_conv_epilogue_emits_layernorm_statistics(dt, kind, tile):
    """theia_gemm_args_t.ln_sums: per-image (sum, sum of squares) of the STORED convolution output, accumulated by the GEMM
    epilogue (all four output-parity launches of a stride-2 transposed convolution add into the same sums; images of 225 / 240
    rows straddle the 128-row wave tiles), and the one-pass LayerNorm[C,H,W] that consumes them == the three-pass one."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    _skip_f32_320(dt, tile)
    b, C = 5, 64
    IH = {"conv_p1": 16, "convT_s1": 14, "convT_s2_p1": 16, "convT_s2_op1": 31}[kind]
    x = h((b, IH, IH, C), 21, 1.0)
    W = h((C, C, 3, 3), 22, 1.0 / math.sqrt(9 * C))
    bias = h((C,), 23, 0.1)
    if kind == "conv_p1":
        plan = ops.plan_conv3x3(C, IH)
    else:
        s, p, op = {"convT_s1": (1, 0, 0), "convT_s2_p1": (2, 1, 0), "convT_s2_op1": (2, 0, 1)}[kind]
        plan = ops.plan_convT3x3(C, IH, s, p, op)
    OH = plan.out_hw
    xd, wf = x.to(dev, dt), _pack(plan.pack_fwd, W, dt, dev)
    out = torch.zeros(b, OH, OH, C, dtype=dt, device=dev)
    sums = torch.zeros(b, 2, dtype=torch.int64, device=dev)  # 2^-24 fixed point: order-independent integer accumulation
    for rmap, mpi in plan.fwd:
        ops.gemm_nt(xd, wf, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, bias=bias.to(dev), act=Nn.ACT_RELU, tile=tile, ln_sums=sums)
    o64 = out.double().view(b, -1)
    fsum = sums.double() / 2 ** 24
    assert relerr(fsum[:, 0], o64.sum(1)) < 1e-5 and relerr(fsum[:, 1], (o64 * o64).sum(1)) < 1e-5
    again = torch.zeros_like(sums)
    for rmap, mpi in plan.fwd:
        ops.gemm_nt(xd, wf, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, bias=bias.to(dev), act=Nn.ACT_RELU, tile=tile, ln_sums=again)
    assert torch.equal(again, sums)  # bit-reproducible
    E = OH * OH * C
    g = (h((E,), 52, 0.2) + 1.0).to(dev)
    s_ = h((E,), 53, 0.2).to(dev)
    y3, st3 = ops.layernorm_chw_fwd(out.view(b, E), g, s_, 1e-5)
    y1, st1 = ops.layernorm_chw_fwd(out.view(b, E), g, s_, 1e-5, sums=sums)
    assert relerr(st1, st3) < 1e-5
    assert relerr(y1.float(), y3.float()) < (1e-5 if dt == torch.float32 else 8e-3)  # bf16: one output ulp where rstd differs in its last bits


def test_transpose_acc_and_wgrad_finish_layouts():
    """theia_transpose_acc_f32 and theia_wgrad_finish against torch indexing: every weight layout of the hot path (nn.Linear
    [n][c], Conv2d [co][ci][3][3], ConvTranspose2d [ci][co][3][3] in both reduction orders), ragged tile edges, accumulation,
    and the bias partials reduced by the same launch."""
    from theia_amd import ops
    dev = _dev()
    src = h((300, 72), 1).to(dev)
    dst = torch.full((72, 300), 0.5, device=dev)
    ops.transpose_acc(src, dst, 300, 72, True)
    assert torch.equal(dst, src.t() + 0.5)
    ops.transpose_acc(src, dst, 300, 72, False)
    assert torch.equal(dst, src.t().contiguous())
    splits, Nn_, C = 3, 40, 72
    for kslots, (sn, ss, sc), view in ((1, (C, 0, 1), lambda t: t.view(Nn_, C)),
                                       (9, (C * 9, 1, 9), lambda t: t.view(Nn_, C, 9).permute(0, 2, 1)),      # W[n][c][slot]
                                       (9, (9, 1, Nn_ * 9), lambda t: t.view(C, Nn_, 9).permute(1, 2, 0))):   # W[c][n][slot]
        slabs = h((splits, Nn_, kslots, C), 7 + kslots).to(dev)
        want = slabs.sum(0)  # [n][slot][c]
        bpart = h((splits, Nn_), 9).to(dev)
        for acc in (False, True):
            out = torch.full((Nn_ * kslots * C,), 0.25, device=dev)
            bout = torch.full((Nn_,), 0.25, device=dev)
            ops.wgrad_finish(slabs, splits, Nn_, kslots, C, out, sn, ss, sc, acc, (bpart, bout, acc))
            got = view(out).reshape(Nn_, kslots, C)
            assert relerr(got, want + (0.25 if acc else 0.0)) < 1e-6
            assert relerr(bout, bpart.sum(0) + (0.25 if acc else 0.0)) < 1e-6
    # nn.Linear gradients take the row kernel (16-byte reduction, no LDS): real sizes, strided output rows (the fused q/k/v
    # gradient writes [3D][D] slices), and the sizes that must fall back to the tiled kernel (C not a multiple of 4)
    for Nn2, C2, sn2, sp in ((768, 3072, 3072, 7), (2304, 768, 768, 9), (96, 64, 80, 5), (33, 70, 70, 3), (40, 66, 72, 2)):
        slabs = h((sp, Nn2, C2), 31 + sp).to(dev)
        want = slabs[0].clone()
        for k in range(1, sp):
            want += slabs[k]  # the kernel's order: split 0 first
        for acc in (False, True):
            out = torch.full((Nn2, sn2), 0.25, device=dev)
            ops.wgrad_finish(slabs, sp, Nn2, 1, C2, out, sn2, 0, 1, acc)
            assert torch.equal(out[:, :C2], want + 0.25 if acc else want), (Nn2, C2, sn2, acc)
            assert bool((out[:, C2:] == 0.25).all())  # the padding between rows is not touched


def test_fp8_quantize_and_scale_update():
    """theia_quantize_fp8 == torch's float8_e4m3fn rounding of the scaled, saturated values (bit for bit), amax tracking and
    the delayed-scaling update."""
    from theia_amd import ops
    dev = _dev()
    x = torch.cat([h((300, 200), 3, 5.0), torch.tensor([[1000.0, -1000.0, 448.0, -448.0, 0.0, 1e-4, 0.0019, 0.001] * 25])], 0)
    for dt in (torch.float32, torch.bfloat16):
        xr = rnd(x, dt)
        amax = torch.zeros(1, device=dev)
        for sc in (1.0, 0.37, 90.0):
            scale = torch.tensor([sc], device=dev)
            q = ops.quantize_fp8(x.to(dev, dt), scale, amax)
            ref = (xr * sc).clamp(-448, 448).to(torch.float8_e4m3fn)
            assert torch.equal(q.cpu().view(torch.uint8), ref.view(torch.uint8)), (dt, sc)
        assert float(amax) == float(xr.abs().max())
        wide = torch.zeros(301, 264, dtype=dt, device=dev)
        wide[:, 64:] = x.to(dev, dt)
        q = ops.quantize_fp8(wide[:, 64:], torch.ones(1, device=dev))  # strided rows, no amax
        assert torch.equal(q.cpu().view(torch.uint8), xr.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
    amax = torch.tensor([2.0, 0.0, 56.0], device=dev)
    scale, inv = torch.ones(3, device=dev), torch.ones(3, device=dev)
    ops.fp8_update_scales(amax, scale, inv, 1.0)
    assert scale.tolist() == [224.0, 1.0, 8.0] and amax.tolist() == [0.0, 0.0, 0.0]
    assert inv.tolist() == pytest.approx([1 / 224.0, 1.0, 0.125], rel=1e-6)


@pytest.mark.parametrize("M,N,K", [(2600, 768, 768), (513, 264, 128), (256 * 5, 512, 64), (25216, 1152, 384)])
def test_fp8_gemm_against_dequantised_reference(M, N, K):
    """THEIA_FP8 theia_gemm_nt: e4m3 operands, f32 accumulation on the fp8 matrix cores, bf16 epilogues -- against an f32 GEMM
    of the DE-QUANTISED operands (so the only differences are accumulation order and the bf16 output rounding, tolerance 2e-2),
    for the epilogues the engine uses."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    x, w = h((M, K), 1, 2.0), h((N, K), 2, 1.0 / math.sqrt(K))
    bias, res = h((N,), 3, 0.1), h((M, N), 4, 1.0)
    sx, sw = torch.tensor([448.0 / float(x.abs().max())], device=dev), torch.tensor([448.0 / float(w.abs().max())], device=dev)
    x8, w8 = ops.quantize_fp8(x.to(dev), sx), ops.quantize_fp8(w.to(dev), sw)
    inv = (1.0 / sx, 1.0 / sw)
    ref = (x8.float().cpu() @ w8.float().cpu().t()) * float(inv[0]) * float(inv[1])
    # the quantisation itself: within e4m3's 2^-4 relative step of the exact product (sanity, not a kernel property)
    assert relerr(ref, x @ w.t()) < 0.1
    y = ops.linear(x8, w8, bias.to(dev), scale_inv=inv)
    assert y.dtype == torch.bfloat16 and relerr(y.float(), ref + bias) < TOL[torch.bfloat16]
    y = ops.linear(x8, w8, bias.to(dev), resid=res.to(dev, torch.bfloat16), scale_inv=inv)
    assert relerr(y.float(), ref + bias + rnd(res, torch.bfloat16)) < TOL[torch.bfloat16]
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    y = ops.linear(x8, w8, bias.to(dev), act=Nn.ACT_GELU, aux_out=pre, scale_inv=inv)
    assert relerr(pre.float(), ref + bias) < TOL[torch.bfloat16]
    assert relerr(y.float(), torch.nn.functional.gelu(ref + bias)) < TOL[torch.bfloat16]
    with pytest.raises(Nn.TheiaNativeError, match="fp8"):
        ops.linear(x8[:, :48].contiguous(), w8[:, :48].contiguous(), scale_inv=inv)  # K = 48 is not a multiple of 64


@pytest.mark.parametrize("kind", ["conv_p1", "convT_s2_op1"])
def test_fp8_implicit_gemm_convolutions(kind):
    """fp8 operands through the row-map gather (3x3 convolution, stride-2 transposed convolution parity classes + data-gradient)."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    b, C = 3, 128
    IH = {"conv_p1": 16, "convT_s2_op1": 31}[kind]
    x = h((b, IH, IH, C), 21, 1.0)
    W = h((C, C, 3, 3), 22, 1.0 / math.sqrt(9 * C))
    bias = h((C,), 23, 0.1)
    plan = ops.plan_conv3x3(C, IH) if kind == "conv_p1" else ops.plan_convT3x3(C, IH, 2, 0, 1)
    OH = plan.out_hw
    one = torch.ones(1, device=dev)
    sx, sw = 448.0 / float(x.abs().max()), 448.0 / float(W.abs().max())
    x8 = ops.quantize_fp8(x.view(-1, C).to(dev), one * sx).view(b, IH, IH, C)
    wf = _pack(plan.pack_fwd, W, torch.float32, dev).view(C, 9 * C)
    wf8 = ops.quantize_fp8(wf, one * sw)
    xq = x8.float().cpu() / sx
    Wq = (_pack(plan.pack_fwd, W, torch.float32, dev).view(C, 9 * C)).cpu()
    # de-quantised weights back in the reference layout: quantisation is element-wise, so quantise the original tensor
    Wdq = (W * sw).clamp(-448, 448).to(torch.float8_e4m3fn).float() / sw
    ref = O.conv3x3_p1(xq, Wdq, bias) if kind == "conv_p1" else O.convT3x3(xq, Wdq, bias, 2, 0, 1)
    out = torch.empty(b, OH, OH, C, dtype=torch.bfloat16, device=dev)
    inv = (one / sx, one / sw)
    for rmap, mpi in plan.fwd:
        ops.gemm_nt(x8, wf8, out, b * mpi, C, rmap.ntaps * C, rmap, 9 * C, C, bias=bias.to(dev), scale_inv=inv)
    assert relerr(out.float(), ref) < TOL[torch.bfloat16]


def test_layernorm_statistics_with_a_residual_run_on_the_2stage_kernel():
    """ln_sums + resid: the ping-pong kernels' statistics instantiations have no residual prefetch, the library routes such a launch
    to the 128x128 kernel (and refuses an explicit 256x256 request instead of computing something else)."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    dt, b, C = torch.bfloat16, 64, 768   # 64 images x 256 rows, N = 768: the library's own choice would be the ping-pong kernel
    x = h((b, 16, 16, C), 81, 1.0)
    W = h((C, C, 3, 3), 82, 1.0 / math.sqrt(9 * C))
    res = h((b, 16, 16, C), 83, 1.0)
    plan = ops.plan_conv3x3(C, 16)
    (rmap, mpi), = plan.fwd
    xd, wf, rd = x.to(dev, dt), _pack(plan.pack_fwd, W, dt, dev), res.to(dev, dt)
    assert Nn.lib().theia_gemm_nt_tile(b * mpi, C, Nn.BF16) == 256256
    out = torch.zeros(b, 16, 16, C, dtype=dt, device=dev)
    sums = torch.zeros(b, 2, dtype=torch.int64, device=dev)
    kw = dict(resid=rd, ln_sums=sums)
    assert ops.gemm_nt(xd, wf, out, b * mpi, C, 9 * C, rmap, 9 * C, C, plan_only=True, **kw) == 128128
    ops.gemm_nt(xd, wf, out, b * mpi, C, 9 * C, rmap, 9 * C, C, **kw)
    plain = torch.zeros_like(out)
    ops.gemm_nt(xd, wf, plain, b * mpi, C, 9 * C, rmap, 9 * C, C, tile=128128)
    assert relerr(out.float(), plain.float() + rd.float()) < 8e-3
    o64 = out.double().view(b, -1)
    fsum = sums.double() / 2 ** 24
    assert relerr(fsum[:, 0], o64.sum(1)) < 1e-5 and relerr(fsum[:, 1], (o64 * o64).sum(1)) < 1e-5
    for tile in (256256, 256009):
        with pytest.raises(Nn.TheiaNativeError, match="ln_sums together with resid"):
            ops.gemm_nt(xd, wf, out, b * mpi, C, 9 * C, rmap, 9 * C, C, tile=tile, **kw)


@pytest.mark.parametrize("tile", [256256, 320256])
def test_work_conserving_tile_schedule_is_bit_identical_to_the_static_rounds(tile):
    """theia_set_gemm_schedule(1) (ABI v10): the persistent NT kernel draws its tiles from per-XCD queues (scalar atomics on a per-launch
    counter) instead of owning them by workgroup index.  Same tiles, same arithmetic: the outputs must be BIT-IDENTICAL to the static
    schedule -- on one-round launches, on launches of several rounds, with the CU budget squeezed to 40 (every workgroup walks over many
    tiles and the queues run dry at different times), with every epilogue flavour that changes what a tile boundary carries (bias row
    prefetch, residual rows = the draining boundary, GELU + saved pre-activation, GELU'), and when a second stream keeps CUs busy."""
    from theia_amd import ops, _native as Nn
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    cases = [(25216, 768, 768, "resid"), (6000, 3072, 768, "gelu"), (6000, 768, 3072, "plain"), (3000, 3072, 768, "dgelu"), (5000, 1024, 256, "plain")]
    prev = ops.set_gemm_schedule(False)
    cus0 = ops.get_compute_cus()
    try:
        for budget in (0, 40):
            ops.set_compute_cus(budget)
            for (M, N, K, kind) in cases:
                x = (torch.randn(M, K, generator=g)).to(dev, torch.bfloat16)
                w = (torch.randn(N, K, generator=g) * 0.05).to(dev, torch.bfloat16)
                b = torch.randn(N, generator=g).to(dev)
                r = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
                outs = []
                for dyn in (False, True, True):
                    ops.set_gemm_schedule(dyn)
                    assert ops.get_gemm_schedule() == dyn
                    pre = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                    if kind == "resid":
                        y = ops.linear(x, w, b, resid=r, tile=tile)
                    elif kind == "gelu":
                        y = ops.linear(x, w, b, act=Nn.ACT_GELU, aux_out=pre, tile=tile)
                    elif kind == "dgelu":
                        y = ops.linear(x, w, None, act=Nn.ACT_MUL_DGELU, aux_in=r, tile=tile)
                    else:
                        y = ops.linear(x, w, b, tile=tile)
                    outs.append((y.clone(), pre.clone()))
                torch.cuda.synchronize()
                for (y, pre) in outs[1:]:
                    assert torch.equal(y, outs[0][0]) and torch.equal(pre, outs[0][1]), (budget, M, N, K, kind)
        # a launch that does not have the chip to itself: a long GEMM on a second stream holds CUs while the dynamic launches run
        ops.set_compute_cus(0)
        M, N, K = 25216, 3072, 768
        x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev, torch.bfloat16)
        b = torch.randn(N, generator=g).to(dev)
        ops.set_gemm_schedule(False)
        ref = ops.linear(x, w, b, tile=tile).clone()
        big_x = torch.randn(32768, 3072, generator=g).to(dev, torch.bfloat16)
        big_w = (torch.randn(768, 3072, generator=g) * 0.05).to(dev, torch.bfloat16)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        ops.set_gemm_schedule(True)
        for rep in range(3):
            with torch.cuda.stream(side):
                ops.linear(big_x, big_w, None, tile=256256)
            y = ops.linear(x, w, b, tile=tile)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), rep
    finally:
        ops.set_gemm_schedule(prev)
        ops.set_compute_cus(cus0 if cus0 != ops.device_cus() else 0)
