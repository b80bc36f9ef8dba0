"""CPU, world_size = 2 over gloo: the data-parallel path (parameter broadcast, bucket-wise gradient averaging in
backward-completion order).  Per-rank gradients come from the CPU oracle on each rank's shard of the batch; the
averaged result must equal the single-process gradient of the whole batch (golden G9, produced by the reference's
own DDP run)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, golden_path, ret):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import theia_oracle as O
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.parallel import GradBucketReducer, TheiaDataParallel, broadcast_parameters

    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["cdiv"]
    torch.manual_seed(100 + rank)  # different random init per rank: the broadcast must fix that
    model = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers})
    if rank == 0:
        model.load_state_dict(O.synth_params(bb, teachers, 0))
    ddp = TheiaDataParallel(model)  # broadcasts rank 0's parameters
    assert ddp.module is model and model.engine.bucket_ready_hook is not None
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = O.synth_params(bb, teachers, 0)
    assert all(torch.equal(params[k], ref[k]) for k in ref), "parameter broadcast failed"

    B = 4
    images = O.synth_images(B, 0)[rank * 2:(rank + 1) * 2]
    targets = {t: v[rank * 2:(rank + 1) * 2] for t, v in O.synth_targets(B, teachers, 1).items()}
    _, main, grads, _ = O.train_step_grads(params, images, targets, bb, teachers, "cos_l1")
    # write the oracle gradients into the engine's flat buckets (what the HIP backward does on the GPU) and reduce
    # bucket by bucket in backward-completion order
    red = ddp.reducer
    assert isinstance(red, GradBucketReducer) and red.world == 2
    name_of = {id(p): n for n, p in model.named_parameters()}
    for b in model.engine.buckets:
        b.ensure(torch.device("cpu"))
        for i, p in enumerate(b.params):
            b.view(i).copy_(grads[name_of[id(p)]])
        red.bucket_ready(b.flat)
    red.finish()
    if rank == 0:
        out = {}
        for b in model.engine.buckets:
            for i, p in enumerate(b.params):
                out[name_of[id(p)]] = float(b.view(i).double().pow(2).sum().sqrt())
        ret["gn"] = out
        ret["main"] = float(main)
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_gloo_bucket_reduction_matches_single_process(golden_dir):
    g = np.load(os.path.join(golden_dir, "g9_dp2_vs_single.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), golden_dir, ret), nprocs=2, join=True)
    gn = ret["gn"]
    assert abs(ret["main"] - float(g["main_rank0_b2"])) < 1e-5 * abs(float(g["main_rank0_b2"]))
    for i, k in enumerate(str(n) for n in g["grad_names"]):
        if "k_proj.bias" in k:
            continue
        ref = float(g["gradnorm_single_b4"][i])
        assert abs(gn[k] - ref) < 1e-3 * ref, (k, gn[k], ref)


def _worker_async_order(rank, world, port, ret, exchange="allreduce", comm_dtype="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    from theia_amd.parallel import GradBucketReducer, broadcast_parameters
    red = GradBucketReducer(exchange=exchange, comm_dtype=comm_dtype)
    # bucket sizes: multiples of 8 like the engine's (divisible by the world size), and one that is not (rs_ag: exchanged through a
    # zero-padded staging buffer with the same shard boundaries on every rank -- it used to fall back to an all-reduce silently)
    flats = [torch.full((1000 + 8 * i,), float(rank + 1) * (i + 1)) for i in range(5)] + [torch.full((1001,), float(rank + 1))]
    flats[2][7] = 0.1 * (rank + 1) + 3.0  # a value bf16 cannot hold exactly: the bf16 exchange rounds, the fp32 one must not
    for f in flats:
        red.bucket_ready(f)
    red.finish()
    want = [torch.full_like(f, 1.5 * (i + 1)) for i, f in enumerate(flats[:5])] + [torch.full((1001,), 1.5)]
    want[2][7] = 0.15 + 3.0
    tol = dict(rtol=0, atol=0) if comm_dtype == "fp32" else dict(rtol=8e-3, atol=0)
    ok = all(torch.allclose(f, w, **(tol if comm_dtype == "bf16" else {})) for f, w in zip(flats, want))
    if comm_dtype == "fp32":
        ok = ok and abs(float(flats[2][7]) - 3.15) < 1e-6
    else:
        ok = ok and float(flats[2][7]) != 3.15 and abs(float(flats[2][7]) - 3.15) < 0.02  # went through bf16
    red.finish()  # idempotent
    if exchange == "rs_ag":  # the odd-length bucket took the padded reduce-scatter + all-gather path, the others ran in place
        ok = ok and len(red._pad) == 1 and next(iter(red._pad.values())).numel() == 1002
    else:
        ok = ok and len(red._pad) == 0
    red.close()  # (no C-ABI communicator on gloo: a no-op, idempotent)
    # coalesced parameter broadcast: mixed shapes / dtypes, more bytes than one flat buffer
    torch.manual_seed(7 + rank)
    ps = [torch.randn(33, 5), torch.randn(7), torch.randn(4, 4, 4).double(), torch.randn(1000), torch.randn(3)]
    broadcast_parameters(ps, 0, bucket_bytes=2048)
    torch.manual_seed(7)
    ref = [torch.randn(33, 5), torch.randn(7), torch.randn(4, 4, 4).double(), torch.randn(1000), torch.randn(3)]
    ok = ok and all(torch.equal(a, b) for a, b in zip(ps, ref))
    if rank == 0:
        ret["ok"] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,comm_dtype", [("allreduce", "fp32"), ("rs_ag", "fp32"), ("allreduce", "bf16"), ("rs_ag", "bf16")])
def test_reducer_averages_several_buckets_async(exchange, comm_dtype):
    """every exchange form of GradBucketReducer (one all-reduce per bucket / reduce-scatter + all-gather on the flat buffer, fp32 / bf16
    on the wire) + the coalesced parameter broadcast, world size 2"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_async_order, args=(2, _free_port(), ret, exchange, comm_dtype), nprocs=2, join=True)
    assert ret["ok"]


def test_rccl_channel_cap_knob(monkeypatch):
    from theia_amd.parallel import DEFAULT_RCCL_CHANNELS, configure_rccl_env, rccl_channels, reserved_cus
    for v in ("NCCL_MAX_NCHANNELS", "THEIA_RCCL_MAX_NCHANNELS", "THEIA_DP_RESERVED_CUS"):
        monkeypatch.delenv(v, raising=False)
    configure_rccl_env()  # default: opt-in -- RCCL's own channel choice, no CUs taken from the GEMM planners
    assert DEFAULT_RCCL_CHANNELS == 0 and "NCCL_MAX_NCHANNELS" not in os.environ and reserved_cus() == 0
    monkeypatch.setenv("THEIA_RCCL_MAX_NCHANNELS", "8")
    configure_rccl_env()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "8" and rccl_channels() == 8 and reserved_cus() == 8
    monkeypatch.setenv("THEIA_DP_RESERVED_CUS", "24")
    assert reserved_cus() == 24
    monkeypatch.delenv("THEIA_DP_RESERVED_CUS", raising=False)
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.setenv("THEIA_RCCL_MAX_NCHANNELS", "0")  # RCCL's own defaults, nothing reserved
    configure_rccl_env()
    assert "NCCL_MAX_NCHANNELS" not in os.environ and reserved_cus() == 0
    monkeypatch.delenv("THEIA_RCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "12")       # a cap the user exported is respected and reserved for
    configure_rccl_env()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "12" and reserved_cus() == 12
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)


def test_single_process_reducer_is_a_noop():
    from theia_amd.parallel import GradBucketReducer
    r = GradBucketReducer()
    f = torch.ones(16)
    r.bucket_ready(f)
    r.finish()
    assert r.world == 1 and torch.equal(f, torch.ones(16))


def test_cu_budget_is_restored_to_what_it_was_not_to_the_whole_device():
    """TheiaDataParallel shrinks the GEMM planners' CU budget while buckets are exchanged; afterwards -- and at the next forward() if a
    backward pass raised before its completion callback -- the budget that was in force BEFORE comes back (a THEIA_COMPUTE_CUS the
    user set survives), not "whole device"."""
    from theia_amd import ops, parallel
    calls = []
    state = {"cus": 200}

    class _Eng:
        bucket_ready_hook = None

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(4))
            self.engine = _Eng()

        def forward(self, x):
            return x

    orig = (ops.get_compute_cus, ops.set_compute_cus)
    ops.get_compute_cus = lambda: state["cus"]
    ops.set_compute_cus = lambda n: (calls.append(n), state.__setitem__("cus", n))
    try:
        ddp = parallel.TheiaDataParallel(_M(), broadcast=False)
        ddp._reserve = 16
        ddp._shrink_cus()
        ddp._shrink_cus()  # idempotent: the saved budget is not overwritten by the reduced one
        assert state["cus"] == 184 and ddp._saved_cus == 200
        ddp._finalize()
        assert state["cus"] == 200 and ddp._saved_cus is None
        ddp._shrink_cus()
        ddp._callback_queued = True  # a backward pass that raised: _finalize never ran
        ddp(torch.zeros(1))
        assert state["cus"] == 200 and not ddp._callback_queued
    finally:
        ops.get_compute_cus, ops.set_compute_cus = orig


def test_cu_reservation_choice_needs_a_clear_gain():
    """TheiaDataParallel.autotune_reserved_cus times a step per candidate reservation; pick_reservation keeps the smallest candidate
    (nothing reserved) unless another one is faster by more than the margin -- noise does not switch a reservation on -- and the
    cheapest of equally fast ones otherwise."""
    from theia_amd.parallel import pick_reservation
    assert pick_reservation({0: 50.0, 16: 49.8, 32: 49.9, 64: 51.0}) == 0      # 0.4 %: noise
    assert pick_reservation({0: 50.0, 16: 47.0, 32: 46.0, 64: 48.0}) == 32     # clear winner
    assert pick_reservation({0: 50.0, 16: 46.0, 32: 46.0, 64: 48.0}) == 16     # tie: the smaller reservation
    assert pick_reservation({0: 50.0}) == 0
    assert pick_reservation({16: 50.0, 32: 45.0}, min_gain=0.05) == 32          # base = the smallest candidate offered
