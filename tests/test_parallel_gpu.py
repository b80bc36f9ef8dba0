"""GPU, world_size = 2: the real data-parallel step -- HIP forward and backward of each rank's shard under
TheiaDataParallel, bucket all-reduces issued on the reducer's stream behind the main-stream and weight-gradient-stream
events, joined by the autograd completion callback -- must give every rank the single-process gradient of the whole batch
(the oracle's, cf. golden G9 for the reference's own DDP run).

  * RCCL, one rank per GPU: runs whenever >= 2 GPUs are visible (the call site replaced: reference train_rvfm.py:211-218,258);
    also `bench.py --gpus 2` launching itself.
  * two ranks on ONE GPU over gloo (RCCL refuses two ranks per GPU): opt-in, the rig is flaky (see below)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

# Opt-in (THEIA_TEST_DP2_ONE_GPU=1): gloo staging device tensors through the host with two processes sharing one GPU
# dead-locks sporadically inside gloo's all_reduce on this stack (seen in bench.py's THEIA_BENCH_ONE_DEVICE smoke run, with
# every rank parked in the same collective), so the test is kept out of the default GPU tier; the N>1 logic itself is
# covered on CPU by tests/test_parallel_gloo.py and the RCCL path by bench.py under torch.distributed.run.
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
one_gpu_rig = pytest.mark.skipif(os.environ.get("THEIA_TEST_DP2_ONE_GPU") != "1", reason="opt-in: flaky gloo-on-one-GPU rig")
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
two_gpus = pytest.mark.skipif(NGPU < 2, reason=f"needs >= 2 GPUs for RCCL (one rank per GPU); {NGPU} visible")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret, backend="gloo", exchange="allreduce", comm_dtype="fp32"):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["THEIA_DP_EXCHANGE"] = exchange
    os.environ["THEIA_DP_COMM_DTYPE"] = comm_dtype
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import theia_oracle as O  # checker only
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.parallel import TheiaDataParallel

    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["cdiv"]
    torch.manual_seed(100 + rank)  # different random init per rank: the broadcast must fix that
    model = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers}, precision="fp32")
    if rank == 0:
        model.load_state_dict(O.synth_params(bb, teachers, 0))
    model = model.to(dev)
    ddp = TheiaDataParallel(model)
    B, per = 4, 2
    images = O.synth_images(B, 0)
    targets = O.synth_targets(B, teachers, 1)
    sl = slice(rank * per, (rank + 1) * per)
    for _ in range(2):  # second pass: accumulate=False overwrite + a second round of all-reduces
        for p in model.parameters():
            p.grad = None
        losses = model.get_loss(ddp(images[sl]), {t: v[sl].to(dev) for t, v in targets.items()}, as_float=False)
        (0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]).backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    if backend == "nccl":  # the start-up measurement of the CU reservation: collective, same decision on every rank

        def one_step():
            ls = model.get_loss(ddp(images[sl]), {t: v[sl].to(dev) for t, v in targets.items()}, as_float=False)
            (0.9 * ls["cos_loss"] + 0.1 * ls["l1_loss"]).backward()

        tune = ddp.autotune_reserved_cus(one_step, candidates=(0, 16), steps=1)
        if rank == 0:
            ret["tune"] = (sorted(tune), ddp._reserve)
    if rank == 0:
        params = O.synth_params(bb, teachers, 0)
        _, _, ref, _ = O.train_step_grads(params, images, targets, bb, teachers, "cos_l1")
        worst = ("", 0.0)
        for n, g in got.items():
            if "k_proj.bias" in n:  # analytically zero gradient
                continue
            r = ref[n]
            e = float((g - r).norm() / (r.norm() + 1e-12))
            if e > worst[1]:
                worst = (n, e)
        ret["worst"] = worst
        ret["n"] = len(got)
        ret["backend"] = dist.get_backend()
        ret["world"] = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()


@two_gpus
@pytest.mark.parametrize("exchange,comm_dtype", [("allreduce", "fp32"), ("rs_ag", "fp32"), ("allreduce", "bf16"), ("rs_ag", "bf16")])
def test_dp2_rccl_matches_single_process_gradient(exchange, comm_dtype):
    """every exchange variant of GradBucketReducer over RCCL (runs where >= 2 GPUs are visible): the in-place reduce_scatter_tensor into
    the rank's own shard + all-gather, and the bf16 wire with the AVG reduction, are RCCL-only branches (gloo substitutes them)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, "nccl", exchange, comm_dtype), nprocs=2, join=True)
    assert ret["backend"] == "nccl" and ret["world"] == 2
    assert ret["n"] > 100
    assert ret["tune"][0] == [0, 16] and ret["tune"][1] in (0, 16)
    name, err = ret["worst"]
    assert err < (2e-4 if comm_dtype == "fp32" else 1e-2), (name, err)


@two_gpus
def test_bench_launches_itself_on_two_gpus():
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--no-roofline", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["value"] > 0 and line["config"]["global_batch"] == 32


def test_abi_communicator_world_1():
    """theia_comm_* (include/theia_hip.h) on the one GPU that is always there: RCCL is bound by dlopen (the copy torch.distributed already
    loaded), a 1-rank communicator is built from a fresh id, and the in-place collectives run on the caller's stream -- mean and sum over
    one rank and a broadcast from rank 0 leave the buffer bit-identical.  GradBucketReducer(backend="abi") routes a bucket through it, with
    and without the bf16 wire.  (N > 1 ranks: test_dp2_abi_backend_matches_single_process_gradient, where >= 2 GPUs are visible.)"""
    import ctypes

    from theia_amd import _native as N
    from theia_amd.parallel import AbiCommunicator, GradBucketReducer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    comm = AbiCommunicator(device=dev)
    assert comm.world == 1 and comm.rank == 0
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    assert N.lib().theia_comm_size(comm._handle, ctypes.byref(w), ctypes.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(1 << 20, generator=g).to(dev, dt)
        want = x.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # enqueued on the CURRENT stream
            comm.allreduce(x, average=True)
            comm.allreduce(x, average=False)
            comm.broadcast(x, 0)
        torch.cuda.current_stream(dev).wait_stream(side)
        assert torch.equal(x, want)
    with pytest.raises(ValueError):
        comm.allreduce(torch.zeros(4, device=dev, dtype=torch.float16))
    for wire in ("fp32", "bf16"):
        red = GradBucketReducer(backend="abi", comm_dtype=wire)
        red._abi = comm
        flat = torch.randn(4096, generator=g).to(dev)
        want = flat.clone() if wire == "fp32" else flat.bfloat16().float()
        work, post = red._reduce(flat, True)
        assert work is None
        if post is not None:
            post()
        assert torch.equal(flat, want)
    comm.close()
    comm.close()  # idempotent


def test_reservation_autotune_is_a_noop_for_one_rank():
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.parallel import TheiaDataParallel
    model = RobotVisionFM(backbone="facebook/deit-tiny-patch16-224", translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes={"facebook/dinov2-large": get_model_feature_size("facebook/dinov2-large", keep_spatial=True)},
                          precision="fp32").to("cuda:0")
    calls = []
    assert TheiaDataParallel(model).autotune_reserved_cus(lambda: calls.append(1)) == {} and not calls


@two_gpus
def test_dp2_abi_backend_matches_single_process_gradient():
    """the data-parallel step with the gradient buckets exchanged by theia_comm_allreduce (THEIA_DP_BACKEND=abi) instead of torch.distributed"""
    mgr = mp.Manager()
    ret = mgr.dict()
    os.environ["THEIA_DP_BACKEND"] = "abi"
    try:
        mp.spawn(_worker, args=(2, _free_port(), ret, "nccl", "allreduce", "fp32"), nprocs=2, join=True)
    finally:
        os.environ.pop("THEIA_DP_BACKEND", None)
    assert ret["n"] > 100
    name, err = ret["worst"]
    assert err < 2e-4, (name, err)


@one_gpu_rig
def test_dp2_on_one_gpu_matches_single_process_gradient():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["n"] > 100
    name, err = ret["worst"]
    assert err < 2e-4, (name, err)


def _captured_worker(rank, world, port, ret, backend="gloo"):
    """world-size-2 job on the captured step (round 6): every rank runs CapturedTrainStep(TheiaDataParallel(model), FusedAdamW) -- graph A,
    the eager bucket exchange through the real GradBucketReducer, graph B -- for 5 steps (2 eager warm-up calls, capture, 3 replays) on its
    half of the batch; rank 0 compares the loss trajectory (mean over the ranks) with a single-process eager loop on the whole batch."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import theia_oracle as O  # checker only (synthetic inputs)
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.parallel import TheiaDataParallel
    from theia_amd.train_graph import CapturedTrainStep

    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    bb, teachers = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"]

    def build():
        m = RobotVisionFM(backbone=bb, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in teachers}, precision="fp32")
        m.load_state_dict(O.synth_params(bb, teachers, 0))
        return m.to(dev)

    model = build()
    ddp = TheiaDataParallel(model)
    opt = FusedAdamW(ddp, lr=1e-3, weight_decay=0.01)
    step = CapturedTrainStep(ddp, opt, grad_clip=0.5, warmup=2)
    assert step.split and step.reducer is ddp.reducer
    B, per, nsteps = 4, 2, 5
    sl = slice(rank * per, (rank + 1) * per)
    mine = []
    for i in range(nsteps):
        images = O.synth_images(B, i)
        targets = O.synth_targets(B, teachers, 100 + i)
        mine.append(float(step(images[sl].to(dev), {t: v[sl].to(dev) for t, v in targets.items()})["main_loss"]))
    torch.cuda.synchronize()
    both = torch.tensor(mine, dtype=torch.float64)
    dist.all_reduce(both)  # (every loss term is a batch mean: the mean of the ranks' losses is the whole-batch loss)
    both /= world
    if rank == 0:
        ref = build()
        oref = FusedAdamW(ref, lr=1e-3, weight_decay=0.01)
        want = []
        for i in range(nsteps):
            images = O.synth_images(B, i).to(dev)
            targets = {t: v.to(dev) for t, v in O.synth_targets(B, teachers, 100 + i).items()}
            oref.zero_grad(set_to_none=True)
            ls = ref.get_loss(ref(images), targets, as_float=False)
            main = 0.9 * ls["cos_loss"] + 0.1 * ls["l1_loss"]
            main.backward()
            oref.clip_grad_norm_(0.5)
            oref.step()
            want.append(float(main))
        torch.cuda.synchronize()
        # Parameters are compared through the losses they produce: Adam's first steps move every weight by ~lr whatever the size of its
        # gradient, so entries whose gradient is rounding noise (dead ReLU regions) walk off in either direction -- 2e-2 of a tensor's
        # scale after 5 steps, in the eager dp tests just as here -- while every step's loss is a function of all parameters
        ret["worst"] = max(abs(a - b) / abs(b) for a, b in zip(both.tolist(), want))
        ret["losses"] = (both.tolist(), want)
        ret["replays"] = step.replays
    dist.barrier()
    dist.destroy_process_group()


@one_gpu_rig
def test_captured_halves_with_the_gradient_exchange_two_ranks_on_one_gpu():
    """the world-size > 1 form of the captured step, executed: two ranks (sharing the one GPU, gloo) x 2 images against one process x 4
    images, 5 AdamW steps with clipping -- the mean of the two half-batch gradients is the whole-batch gradient (golden G9), so the
    parameters agree to fp32 rounding amplified by Adam (same bound as the eager dp2 tests use for gradients, per step)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_captured_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    print(f"[captured halves, world 2] replays {ret['replays']}, losses (mean of the ranks) {[round(v, 6) for v in ret['losses'][0]]} vs one "
          f"process {[round(v, 6) for v in ret['losses'][1]]}: worst relative deviation {ret['worst']:.2e}")
    assert ret["replays"] == 3 and ret["worst"] < 1e-4 and ret["losses"][1][-1] < ret["losses"][1][0], dict(ret)


@two_gpus
def test_captured_halves_with_the_gradient_exchange_rccl():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_captured_worker, args=(2, _free_port(), ret, "nccl"), nprocs=2, join=True)
    assert ret["replays"] == 3 and ret["worst"] < 1e-4, dict(ret)
