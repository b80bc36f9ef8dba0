"""CPU: the C-ABI library loads and exports every symbol include/theia_hip.h declares; argument validation of the
entry points (no GPU work is launched); host-side logic (row maps, buckets, decay rule, input handling)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "theia_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(theia_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from theia_amd import _native as N
    lib = N.lib()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/theia_hip.h but not exported"
    assert sorted(N.EXPORTED_SYMBOLS) == declared, "ctypes signature table and header disagree"
    assert lib.theia_abi_version() == N.ABI_VERSION == 12
    assert lib.theia_dtype_size(N.F32) == 4 and lib.theia_dtype_size(N.BF16) == 2 and lib.theia_dtype_size(7) == -1


def test_quantize_batch_plan_and_group_split_count_are_host_logic():
    """theia_quantize_fp8_batch_plan (ABI v11): block prefix sums of a host job table, bad tables refused; theia_wgrad_group_splits: CUs / tiles"""
    from theia_amd import _native as N
    lib = N.lib()
    jobs = (N.QuantJob * 3)()
    for j, n in zip(jobs, (8192, 8200, 64)):
        j.src, j.dst, j.scale, j.n = 1024, 2048, 4096, n
    assert lib.theia_quantize_fp8_batch_plan(C.addressof(jobs), 3) == 1 + 2 + 1
    assert [j.first_block for j in jobs] == [0, 1, 3]
    jobs[1].n = 12  # not a multiple of 8
    assert lib.theia_quantize_fp8_batch_plan(C.addressof(jobs), 3) < 0
    cus = lib.theia_get_compute_cus()
    assert lib.theia_wgrad_group_splits(25216, 36) == cus // 36 and lib.theia_wgrad_group_splits(64, 1) == 1


def test_gemm_schedule_switch_is_host_state():
    """theia_set_gemm_schedule / theia_get_gemm_schedule (ABI v10): host-side state of the persistent NT GEMM's tile schedule, static by
    default, returns the previous setting (no launch involved)."""
    from theia_amd import ops
    first = ops.get_gemm_schedule()
    assert first == (os.environ.get("THEIA_PP_DYNAMIC", "0") not in ("0", ""))
    assert ops.set_gemm_schedule(True) == first and ops.get_gemm_schedule() is True
    assert ops.set_gemm_schedule(False) is True and ops.get_gemm_schedule() is False
    ops.set_gemm_schedule(first)


def test_struct_layout_matches_header(tmp_path):
    """The ctypes mirrors against the C compiler's layout of include/theia_hip.h (sizeof + offsets of the fields the kernels' callers
    set last, which move whenever anything before them does)."""
    import subprocess
    from theia_amd import _native as N
    probes = [("theia_rowmap_t", N.RowMap, "in_batch_stride"), ("theia_gemm_args_t", N.GemmArgs, "tile"),
              ("theia_gemm_args_t", N.GemmArgs, "w_scale_inv"), ("theia_gemm_args_t", N.GemmArgs, "out8_scale"), ("theia_wgrad_args_t", N.WgradArgs, "defer_bias_reduce"),
              ("theia_cast_job_t", N.CastJob, "first_block"), ("theia_quant_job_t", N.QuantJob, "first_block"), ("theia_q8_out_t", N.Q8Out, "amax"),
              ("theia_wgrad_finish_job_t", N.WgradFinishJob, "bias_accumulate")]
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"theia_hip.h\"\nint main(void){\n"
    for cname, _, field in probes:
        src += f'printf("%zu %zu\\n", sizeof({cname}), offsetof({cname}, {field}));\n'
    src += "return 0;}\n"
    (tmp_path / "p.c").write_text(src)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(tmp_path / "p.c"), "-o", str(tmp_path / "p")], check=True)
    rows = subprocess.run([str(tmp_path / "p")], check=True, capture_output=True, text=True).stdout.split("\n")
    for (cname, ct, field), row in zip(probes, rows):
        size, off = map(int, row.split())
        assert C.sizeof(ct) == size, cname
        assert getattr(ct, field).offset == off, f"{cname}.{field}"
    # theia_rowmap_t: 40 int32 then 4 int64; theia_gemm_args_t: 8 pointers, 7 int32 + pad, map, tile + reserved, 3 pointers
    assert C.sizeof(N.RowMap) == 160 + 32
    assert C.sizeof(N.GemmArgs) == 8 * 8 + 4 * 7 + 4 + C.sizeof(N.RowMap) + 8 + 5 * 8  # (+ out8, out8_scale: ABI v11)


def test_host_side_planning_functions():
    """Pure host entry points of the library (no launch): cast-batch scheduling, weight-gradient splits, bias fusion query."""
    from theia_amd import _native as N
    lib = N.lib()
    assert C.sizeof(N.CastJob) == 2 * 8 + 4 * 4 + 5 * 8 + 8 + 4 * 4
    jobs = (N.CastJob * 4)()
    shapes = [(1, 100, 72, 0, 72, 1), (1, 72, 100, 0, 1, 72), (48, 9, 48, 9 * 48, 1, 9), (1, 1, 77, 0, 77, 1)]
    for j, (d0, d1, d2, s0, s1, s2) in zip(jobs, shapes):
        j.src, j.dst, j.d0, j.d1, j.d2, j.s0, j.s1, j.s2, j.t0, j.t1 = 4096, 8192, d0, d1, d2, s0, s1, s2, d1 * d2, d2
    total = lib.theia_cast_batch_plan(C.addressof(jobs), 4)
    blocks = 0
    for j, (d0, d1, d2, *_s) in zip(jobs, shapes):
        assert j.tile1 * j.tile2 == 4096 and j.tile2 % 4 == 0 and j.first_block == blocks
        assert j.tiles1 == -(-d1 // j.tile1) and j.tiles2 == -(-d2 // j.tile2)
        blocks += d0 * j.tiles1 * j.tiles2
    assert total == blocks
    assert [j.tile1 for j in jobs] == [64, 64, 16, 1]
    jobs[1].d1 = 0
    assert lib.theia_cast_batch_plan(C.addressof(jobs), 4) == -1
    # weight-gradient planning: the ping-pong kernel fills one round of 256 CUs; small shapes fall back
    assert lib.theia_wgrad_splits(25216, 768, 768) == 28 and lib.theia_wgrad_splits(25216, 3072, 768) == 7
    assert lib.theia_wgrad_splits(32768, 768, 6912) == 3 and lib.theia_wgrad_splits(128, 1024, 768) == 1
    w = N.WgradArgs()
    w.N, w.map.in_c = 768, 768
    assert lib.theia_wgrad_fuses_bias(w, N.BF16) == 1 and lib.theia_wgrad_fuses_bias(w, N.F32) == 0
    w.map.in_c = 64
    assert lib.theia_wgrad_fuses_bias(w, N.BF16) == 0
    assert lib.theia_gemm_nt_tile(25216, 768, N.BF16) == 256256 and lib.theia_gemm_nt_tile(100, 64, N.BF16) in (128064, 128128)


def test_compute_cu_budget_steers_the_planners():
    """theia_set_compute_cus (host state, no GPU): the weight-gradient planner fills the budget, not the device; the tile-height choice
    of the persistent NT kernel follows it; 0 restores the whole device (256 CUs when no device is visible)."""
    from theia_amd import _native as N, ops
    lib = N.lib()
    try:
        ops.set_compute_cus(0)
        full = ops.get_compute_cus()
        assert full == ops.device_cus() and full >= 64
        if full != 256:
            pytest.skip("planner expectations below are written for 256 CUs")
        b, C = 128, 768
        assert ops.wgrad_splits(b * 197, C, 4 * C) == 7 and ops.wgrad_splits(b * 256, C, 9 * C) == 3 and ops.wgrad_splits(b * 197, C, C) == 28
        ops.set_compute_cus(240)
        assert ops.get_compute_cus() == 240
        # 36 / 81 / 9 tiles: one round of 240 CUs
        assert ops.wgrad_splits(b * 197, C, 4 * C) == 6 and ops.wgrad_splits(b * 256, C, 9 * C) == 2 and ops.wgrad_splits(b * 197, C, C) == 26
        g = N.GemmArgs()
        g.M, g.N, g.K, g.act = b * 197, C, C, N.ACT_NONE
        g.map = ops.rm_plain(C, C, C)
        assert lib.theia_gemm_nt_plan(g, N.BF16) == 320256          # 237 tiles of 320 rows: still one round of 240
        ops.set_compute_cus(200)
        assert lib.theia_gemm_nt_plan(g, N.BF16) == 256256          # 2 rounds either way: 2 x 256 < 2 x 320
        ops.set_compute_cus(100000)
        assert ops.get_compute_cus() == full                       # clamped to the device
        with pytest.raises(RuntimeError):
            ops.set_compute_cus(-1)
    finally:
        ops.set_compute_cus(0)


def test_wgrad_row_mode_of_a_launch_is_host_logic():
    """theia_gemm_wgrad_plan (no launch, no GPU): the 2-stage kernel for shapes the ping-pong kernel does not take, and for the ping-pong
    kernel which row addressing -- plain matrices (every nn.Linear), periodic rows (16x16 output maps over a dense 16x16 or a 14x14
    input, 8x8 maps), stepped rows (the stride-2 transposed convolutions, reduced over 31x31 / 16x16 input pixels with a non-dividing
    width; images of exactly 32 pixels)."""
    from theia_amd import _native as N, ops
    lib = N.lib()
    C, b = 768, 128

    def plan(M, Nn, rmap, ldo, kslots=1, dtype=N.BF16):
        g = N.WgradArgs()
        g.M, g.N, g.ldo, g.kslots, g.splits = M, Nn, ldo, kslots, 1
        g.map = rmap
        return lib.theia_gemm_wgrad_plan(g, dtype)

    assert plan(b * 197, C, ops.rm_plain(C, C, C), C) == 111 and plan(b * 197, 3 * C, ops.rm_plain(C, C, 3 * C), 3 * C) == 111
    assert plan(b * 197, C, ops.rm_plain(C, C, C), C, dtype=N.F32) == 0          # exact-f32 mode: the 2-stage kernel
    assert plan(b * 256, 32, ops.rm_plain(C, C, 32), 32) == 0                    # N < 128
    assert plan(b * 4096, 32, ops.rm_plain(C, C, 32), 32) == 111                 # ... unless M is huge (round 6: the Depth head, read-bound)
    assert lib.theia_wgrad_splits(b * 4096, 32, C) == 128                        # 2 tiles of 128 (n) x 384 (c): a split per CU
    assert plan(b * 197, C, ops.rm_plain(96, 96, C), C) == 0                     # in_c < 128
    # partial c tiles (round 4): in_c = 384 (DeiT-small) / 192 (DeiT-tiny) run the ping-pong kernel too
    assert plan(b * 197, C, ops.rm_plain(192, 192, C), C) == 111 and plan(b * 197, 384, ops.rm_plain(384, 384, 384), 384) == 111
    assert plan(b * 256, 384, ops.plan_conv3x3(384, 16).fwd[0][0], 384, 9) == 112 and plan(b * 256, 192, ops.plan_conv3x3(192, 16).fwd[0][0], 192, 9) == 112
    # round 6: N = in_c = 384 is cut into 128 (n) x 384 (c) tiles -- 3 per tap, all of them full -- instead of 2 x 2 of 256 x 256
    assert lib.theia_wgrad_tiles(384, 384) == 3 and lib.theia_wgrad_tiles(768, 768) == 9 and lib.theia_wgrad_tiles(1152, 384) == 9
    assert lib.theia_wgrad_tiles(192, 192) == 1 and lib.theia_wgrad_tiles(32, 768) == 0
    assert lib.theia_wgrad_splits_taps(b * 256, 384, 9, 384) == 256 // (3 * 9)
    assert lib.theia_wgrad_splits_taps(b * 197, 768, 1, 768) == lib.theia_wgrad_splits(b * 197, 768, 768)
    for p_, want in ((ops.plan_conv3x3(C, 16), 112), (ops.plan_convT3x3(C, 14, 1, 0, 0), 112), (ops.plan_conv3x3(C, 8), 112)):
        assert not p_.wgrad_swapped
        for rmap, mpi in p_.fwd:
            assert plan(b * mpi, C, rmap, C, 9) == want, (want, mpi)
    for p_ in (ops.plan_convT3x3(C, 16, 2, 1, 0), ops.plan_convT3x3(C, 31, 2, 0, 1)):  # reduced over input pixels: 256 (stride-2 gather) / 961
        assert p_.wgrad_swapped
        rmap, mpi = p_.dgrad
        assert plan(b * mpi, C, rmap, C, 9) in (110, 112) and (mpi % 32 != 0) == (plan(b * mpi, C, rmap, C, 9) == 110)
    small = ops.plan_conv3x3(C, 4)  # hypothetical 4x4 maps: 16 pixels per image -> two images per step: stepped rows
    assert plan(b * 16, C, small.fwd[0][0], C, 9) == 110


def test_every_bench_size_gemm_dispatches_the_pingpong_tile():
    """The (M, N) of every theia_gemm_nt launch of the default bench step (DeiT-base + cddsv, per-GPU batch 128; table in
    profiles/*_gemm_shapes_isolated.txt) -> 256x256 (the kernel the forced-tile parity tests cover), except the 32-channel
    Depth-Anything head Linear, which takes the 128x64 tile."""
    from theia_amd import _native as N
    lib = N.lib()
    b = 128
    shapes = [(b * 197, 768), (b * 197, 2304), (b * 197, 3072), (b * 196, 768), (b * 256, 768), (b * 256, 1024), (b * 256, 1280),
              (b * 961, 768), (b * 1024, 768), (b * 240, 768), (b * 225, 768), (b * 4096, 768), (b * 4096, 256)]
    for M, Nn in shapes:
        assert lib.theia_gemm_nt_tile(M, Nn, N.BF16) == 256256, (M, Nn)
    assert lib.theia_gemm_nt_tile(b * 4096, 32, N.BF16) == 128064
    assert lib.theia_gemm_nt_tile(b * 197, 768, N.F32) == 128128  # the exact-f32 mode stays on the 2-stage kernel


def test_resize_plan_tables_are_consistent():
    from theia_amd.preprocess import resample_tables
    for in_size, out_size, rs in ((300, 224, 2), (50, 224, 3), (1000, 224, 2), (224, 256, 3), (7, 224, 2)):
        bounds, w, ksize = resample_tables(in_size, out_size, rs)
        assert bounds.shape == (out_size, 2) and w.shape == (out_size, ksize)
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= in_size).all() and (bounds[:, 1] >= 1).all()
        sums = w.astype(np.int64).sum(1)
        assert np.abs(sums - (1 << 22)).max() <= ksize  # weights sum to 1.0 in 22-bit fixed point up to per-tap rounding
        live = np.arange(ksize)[None, :] < bounds[:, 1:2]
        assert (w[~live] == 0).all()


def test_argument_validation_errors_without_gpu():
    from theia_amd import _native as N
    lib = N.lib()
    g = N.GemmArgs()
    assert lib.theia_gemm_nt(g, N.BF16, None) == -1
    assert b"null operand" in lib.theia_last_error()
    g.a = g.w = g.out = 4096
    g.M, g.N, g.K = 8, 12, 64
    assert lib.theia_gemm_nt(g, N.BF16, None) == -1 and b"multiple of 8" in lib.theia_last_error()
    assert lib.theia_gemm_nt(g, 5, None) == -1 and b"dtype" in lib.theia_last_error()
    assert lib.theia_layernorm_fwd(4096, 4096, 4096, 4096, 4096, 4096, 4, 100, 1e-12, N.F32, None) == -1
    assert lib.theia_attention_fwd(4096, 4096, 4096, 1, 300, 3, N.F32, None) == -1 and b"n=300" in lib.theia_last_error()
    assert lib.theia_distill_loss_fwd(4096, 4096, 4096, 4096, 4096, 2, 12, N.F32, None) == -1
    w = N.WgradArgs()
    assert lib.theia_gemm_wgrad(w, N.BF16, None) == -1
    assert lib.theia_wgrad_splits(25216, 768, 768) >= 1
    with pytest.raises(N.TheiaNativeError):
        N.check(-1, "x")
    # theia_comm_*: argument checks come before RCCL is looked for; a NULL communicator is a no-op to destroy
    assert lib.theia_comm_allreduce(None, None, 0, N.F32, 1, None) == -1 and b"theia_comm_allreduce" in lib.theia_last_error()
    assert lib.theia_comm_broadcast(None, 4096, 16, N.F32, 0, None) == -1
    assert lib.theia_comm_allreduce(4096, 4096, 16, N.FP8, 1, None) == -1 and b"dtype" in lib.theia_last_error()
    assert lib.theia_comm_unique_id(None) == -1 and lib.theia_comm_destroy(None) == 0
    import ctypes
    h = ctypes.c_void_p()
    assert lib.theia_comm_init(ctypes.byref(h), b"\0" * N.COMM_ID_BYTES, 2, 2) == -1 and b"rank 2 of 2" in lib.theia_last_error()


def test_convT_parity_classes_cover_output_exactly_once():
    from theia_amd import ops
    for IH, s, p, op in ((16, 2, 1, 0), (31, 2, 0, 1), (14, 1, 0, 0)):
        plan = ops.plan_convT3x3(64, IH, s, p, op)
        OH = plan.out_hw
        hit = np.zeros((OH, OH), dtype=int)
        taps_total = set()
        for rmap, mpi in plan.fwd:
            assert mpi == rmap.rows_h * rmap.rows_w
            for ry in range(rmap.rows_h):
                for rx in range(rmap.rows_w):
                    hit[ry * rmap.out_sy + rmap.out_y0, rx * rmap.out_sx + rmap.out_x0] += 1
            for t in range(rmap.ntaps):
                assert rmap.wslot[t] not in taps_total
                taps_total.add(rmap.wslot[t])
        assert (hit == 1).all()
        assert taps_total == set(range(9))
        # cross-check tap geometry against the definition out[i*s - p + k] += in[i] * W[k]
        for rmap, _ in plan.fwd:
            for t in range(rmap.ntaps):
                ky, kx = divmod(rmap.wslot[t], 3)
                oy = rmap.out_y0 + 2 * rmap.out_sy if rmap.rows_h > 2 else rmap.out_y0
                ry = (oy - rmap.out_y0) // rmap.out_sy
                i = ry * rmap.in_sy + rmap.dy[t]
                assert i * s - p + ky == oy


def test_input_handling_matches_reference_contract():
    from PIL import Image
    from theia_amd.engine import preprocess_lut, to_uint8_batch
    from oracle import theia_oracle as O
    assert np.array_equal(preprocess_lut(True, True, O.IMAGENET_MEAN, O.IMAGENET_STD), O.preprocess_lut())
    assert np.array_equal(preprocess_lut(False, False, O.IMAGENET_MEAN, O.IMAGENET_STD), np.tile(np.arange(256, dtype=np.float32), (3, 1)))
    img = O.synth_images(2, 1)
    t, cl = to_uint8_batch(img)
    assert cl and t.shape == (2, 224, 224, 3)
    t, cl = to_uint8_batch(img.permute(0, 3, 1, 2))
    assert not cl and t.shape == (2, 3, 224, 224)
    t, cl = to_uint8_batch([Image.fromarray(img[i].numpy()) for i in range(2)])
    assert cl and torch.equal(t, img)
    t, cl = to_uint8_batch(img[0].numpy())
    assert t.shape == (1, 224, 224, 3)
    with pytest.raises(TypeError):
        to_uint8_batch(img.float())
    with pytest.raises(NotImplementedError):
        to_uint8_batch(torch.zeros(1, 256, 256, 3, dtype=torch.uint8))


def test_model_state_dict_buckets_and_decay_rule():
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.optimizers import param_groups_weight_decay
    from oracle import theia_oracle as O
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["cddsv"]
    m = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers})
    shapes = O.param_shapes(bb, teachers)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())  # same names AND registration order as the reference
    for k, s in shapes.items():
        assert tuple(sd[k].shape) == tuple(s), k
    groups = param_groups_weight_decay(m, 0.01)
    nodecay = {id(p) for p in groups[0]["params"]}
    named = dict(m.named_parameters())
    # reference quirk (optimizers/utils.py:30): cls/pos and the 3-D LayerNorm weights DO decay; every *.bias does not
    assert id(named["backbone.model.embeddings.cls_token"]) not in nodecay
    assert id(named["backbone.model.embeddings.position_embeddings"]) not in nodecay
    k3 = "translator.translator_heads.facebook/sam-vit-huge.adapter.6."
    assert id(named[k3 + "weight"]) not in nodecay and id(named[k3 + "bias"]) in nodecay
    assert id(named["backbone.model.layernorm.weight"]) in nodecay
    # buckets: every parameter exactly once, decayed ones first, 32-byte aligned views
    seen = set()
    for b in m.engine.buckets:
        assert b.numel % 8 == 0 and all(o % 8 == 0 for o in b.offsets)
        for i, (n, p) in enumerate(zip(b.names, b.params)):
            assert id(p) not in seen
            seen.add(id(p))
            assert (b.offsets[i] < b.decay_numel) == (id(p) not in nodecay), n
    assert seen == {id(p) for p in m.parameters()}
    assert [b.name for b in m.engine.buckets][5:] == ["vit:9-11", "vit:6-8", "vit:3-5", "vit:2-2", "vit:1-1", "vit:0-0"]
    assert [b.name.split(":")[0] for b in m.engine.buckets] == ["head"] * 5 + ["vit"] * 6
    with pytest.raises(RuntimeError):
        m.forward_feature(O.synth_images(1))  # CPU model: the product path refuses, it never falls back


def test_legacy_checkpoint_key_remap():
    from theia_amd.models.backbones import remap_legacy_key
    assert remap_legacy_key("backbone.model.encoder.layer.3.attention.attention.query.weight") == "backbone.model.layers.3.attention.q_proj.weight"
    assert remap_legacy_key("backbone.model.encoder.layer.0.attention.output.dense.bias") == "backbone.model.layers.0.attention.o_proj.bias"
    assert remap_legacy_key("backbone.model.encoder.layer.11.intermediate.dense.weight") == "backbone.model.layers.11.mlp.fc1.weight"
    assert remap_legacy_key("backbone.model.encoder.layer.11.output.dense.weight") == "backbone.model.layers.11.mlp.fc2.weight"
    assert remap_legacy_key("backbone.model.embeddings.cls_token") == "backbone.model.embeddings.cls_token"


def test_kernel_plan_of_a_launch_is_host_logic():
    """theia_gemm_nt_plan (no launch, no GPU): which kernel a theia_gemm_nt call would run -- the library's own choice for the bench
    shapes, the 3x3 kernel for one-image-per-tile convolutions, explicit requests honoured or refused (never replaced), and the
    LayerNorm-statistics epilogue together with a residual routed to the 2-stage kernel."""
    from theia_amd import _native as N, ops
    lib = N.lib()

    def plan(M, Nn, K, rmap, dtype=N.BF16, tile=0, ln_sums=0, resid=0, act=N.ACT_NONE, aux_in=0, rowtab=0):
        g = N.GemmArgs()
        g.rowtab, g.rowtab_period = rowtab or None, 196
        g.a, g.w, g.out = 4096, 8192, 12288  # never dereferenced by the planner
        g.M, g.N, g.K, g.ldw, g.ldo, g.act, g.tile = M, Nn, K, K, Nn, act, tile
        g.map = rmap
        g.ln_sums, g.resid, g.aux_in = ln_sums or None, resid or None, aux_in or None
        return lib.theia_gemm_nt_plan(g, dtype)

    b, C = 128, 768
    lin = ops.rm_plain(C, C, C)
    # 25216 rows x 768 columns: 79 tiles of 320 rows x 3 = ONE round of the persistent grid (297 tiles of 256 rows would be two)
    assert plan(b * 197, C, C, lin) == 320256 and plan(b * 197, C, C, lin, N.F32) == 128128
    assert plan(b * 197, 3 * C, C, ops.rm_plain(C, C, 3 * C)) in (256256, 320256) and plan(b * 197, 4 * C, C, ops.rm_plain(C, C, 4 * C)) == 256256
    assert plan(64, C, C, lin) in (128128, 128064) and plan(64, C, C, lin, tile=256256) == 256256 and plan(64, C, C, lin, tile=320256) == 320256
    assert plan(64, C, C, lin, N.F32, tile=320256) < 0                                        # 320-row tiles are bf16 only
    # what the persistent ping-pong kernel leaves to the 2-stage kernels: a position row table, residual + activation
    assert plan(b * 196, C, C, lin, rowtab=4096) == 256000 and plan(b * 196, C, C, lin, rowtab=4096, tile=256256) < 0
    assert plan(b * 197, C, C, lin, resid=4096, act=N.ACT_RELU) == 256000 and plan(b * 197, C, C, lin, resid=4096) == 320256
    conv = ops.plan_conv3x3(C, 16)
    (fmap, mpi), = conv.fwd
    assert plan(b * mpi, C, 9 * C, fmap) == 256009 and plan(b * mpi, C, 9 * C, conv.dgrad[0]) == 256009
    assert plan(b * mpi, C, 9 * C, fmap, tile=256256) == 256256 and plan(b * mpi, C, 9 * C, fmap, tile=128128) == 128128
    up = ops.plan_convT3x3(C, 16, 2, 1, 0)      # stride-2 transposed convolution: parity classes, not one image per tile
    assert all(plan(b * m, C, r.ntaps * C, r) in (256256, 320256) for r, m in up.fwd)
    assert plan(b * up.fwd[0][1], C, up.fwd[0][0].ntaps * C, up.fwd[0][0], tile=256009) < 0   # refused, not replaced
    assert plan(b * 197, C, 40, ops.rm_plain(40, 40, C), tile=256256) < 0                     # K not a multiple of 32
    # LayerNorm statistics + a residual / aux_in row: no such instantiation of the ping-pong kernels
    assert plan(b * mpi, C, 9 * C, fmap, ln_sums=4096) == 256009
    assert plan(b * mpi, C, 9 * C, fmap, ln_sums=4096, resid=4096) == 128128
    assert plan(b * mpi, C, 9 * C, fmap, ln_sums=4096, act=N.ACT_MUL_DRELU, aux_in=4096) == 128128
    assert plan(b * mpi, C, 9 * C, fmap, ln_sums=4096, resid=4096, tile=256009) < 0
    assert plan(b * mpi, C, 9 * C, fmap, ln_sums=4096, resid=4096, tile=256256) < 0
    assert b"ln_sums together with resid" in lib.theia_last_error()
