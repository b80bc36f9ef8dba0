#!/usr/bin/env python
"""bench.py -- images/sec of one Theia distillation train step on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N == 1 runs in this process.  N > 1 without WORLD_SIZE in the environment re-launches itself as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``
(one rank per GPU over RCCL); launched that way by someone else (RANK / WORLD_SIZE set) it just runs as a rank.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): DeiT-base-patch16-224 student, 5
teachers (cddsv: ViT-H, DINOv2-L, CLIP-L, SAM-H, Depth-Anything-L), bf16 MFMA operands / f32 accumulate / f32 master
weights, per-GPU batch 128 (global 1024 at 8 GPUs -> weak scaling), loss 0.9*cos + 0.1*smooth-L1.
One step = forward + loss + backward + gradient all-reduce (RCCL, overlapped) + fused AdamW update, on synthetic
uint8 224x224x3 images and random f32 teacher features already resident in HBM.  Random-init weights.
``--mode forward_feature`` measures BASELINE configs[4] instead (see forward_feature_main).

Before anything is timed the run CHECKS ITSELF and refuses to report a value if a check fails:
  1. rank 0: one bf16 step of the same architecture at batch 2 with every GEMM forced onto the 256x256 ping-pong kernel
     (the kernel the batch-128 step runs on) against the CPU oracle's fp32 losses and gradients;
  2. every rank: the step at the bench batch with the library's own dispatch against the same step with every GEMM
     forced onto the 2-stage 128x128 kernel (the one the small-shape oracle tests use by default).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline          dominant kernel = the theia_gemm_nt tile variant with the largest share of the step (MFMA-bound):
                    algorithmic FLOPs of its launches / their HIP-event-measured duration (events on the launch stream,
                    taken in extra instrumented steps right after the timed region so the timed steps stay unperturbed)
  student_roofline  DeiT-base student alone, forward + backward (north_star's ">= 40 % of bf16 MFMA peak" claim):
                    105.147 GFLOP per image / HIP-event time of backbone forward+backward
  cpu_baseline      SURVEY 8(d): the CPU oracle (oracle/theia_oracle.py, torch fp32) running BASELINE configs[0] exactly
                    (DeiT-tiny, dinov2-large, batch 8, 10 warm-up + 50 timed steps, median) on the host's physical cores;
                    rank 0, N == 1 only.  The base+5-teacher oracle step of self-check 1 is reported beside it.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BACKBONE = "facebook/deit-base-patch16-224"
TEACHERS = [
    "google/vit-huge-patch14-224-in21k",
    "facebook/dinov2-large",
    "openai/clip-vit-large-patch14",
    "facebook/sam-vit-huge",
    "LiheYoung/depth-anything-large-hf",
]
TEACHER_SETS = {  # configs/training/target_models/{dinov2,cdiv,cddsv}.yaml of the reference (dict order = head order)
    "dinov2": ["facebook/dinov2-large"],
    "cdiv": ["google/vit-huge-patch14-224-in21k", "facebook/dinov2-large", "openai/clip-vit-large-patch14"],
    "cddsv": list(TEACHERS),
}
FLOPS_PER_IMAGE_FWD_BWD = 272.169e9   # BASELINE.md sec. 3 (2xMAC, fwd+bwd, base + cddsv)
# C1 (tiny + dinov2), C2 (tiny + cdiv), C3 (base + cddsv), C4 (small + cddsv); other combinations: not tabulated
FLOPS_PER_IMAGE = {("facebook/deit-base-patch16-224", "cddsv"): 272.169e9, ("facebook/deit-small-patch16-224", "cddsv"): 71.571e9,
                   ("facebook/deit-tiny-patch16-224", "cdiv"): 12.673e9, ("facebook/deit-tiny-patch16-224", "dinov2"): 9.175e9}
FLOPS_PER_IMAGE_STUDENT = 105.147e9   # DeiT-base backbone only, fwd+bwd
FLOPS_PER_IMAGE_FWD = 35.126e9        # DeiT-base forward
MFMA_BF16_PEAK = 2.5e15               # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12                     # HBM3E, same guide (a copy reaches ~6.3e12)
METRIC = "images/sec train-step (fwd+bwd+allreduce) DeiT-base 5-teacher"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--backbone", default=BACKBONE)
    ap.add_argument("--teachers", default="cddsv", choices=sorted(TEACHER_SETS),
                    help="teacher set (reference configs/training/target_models/*.yaml): cddsv = 5 teachers (BASELINE configs[2..3]), "
                         "cdiv = ViT-H + DINOv2 + CLIP (configs[1]), dinov2 = 1 teacher (configs[0])")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8: BASELINE configs[3] mode (with --backbone facebook/deit-small-patch16-224 --batch 256)")
    ap.add_argument("--mode", default="train", choices=["train", "forward_feature"])
    ap.add_argument("--chunk", type=int, default=512, help="forward_feature: images per captured graph replay")
    ap.add_argument("--stream-batch", type=int, default=4096, help="forward_feature: images streamed per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: run the step as ONE hipGraph replay (theia_amd/train_graph.py) instead of ~1500 eager launches")
    ap.add_argument("--no-selfcheck", action="store_true", help="tuning runs only: the printed line is marked unchecked")
    ap.add_argument("--no-dp-autotune", action="store_true", help="N > 1: skip the start-up measurement of the CU reservation for RCCL")
    ap.add_argument("--no-optimizer", action="store_true", help="exclude the AdamW update from the step")
    ap.add_argument("--teacher-dtype", default="auto", choices=["auto", "bf16", "fp32"],
                    help="dtype of the resident teacher features.  auto = SURVEY.md 8(d): bf16 for the throughput modes (bf16 / fp8 -- the "
                         "reference's features ARE bf16 values, data_utils.py:374-379), fp32 for --precision fp32")
    return ap.parse_args(argv)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ N > 1 launch
def self_spawn(args, argv) -> None:
    """--gpus N > 1 outside a torch.distributed launch: become the launcher (one rank per GPU, RCCL)."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "images/sec", "n_gpus": args.gpus,
                          "error": f"--gpus {args.gpus} but only {ndev} GPU(s) are visible"}), flush=True)
        raise SystemExit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("launching " + " ".join(cmd))
    raise SystemExit(subprocess.call(cmd))


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cpu():
    """(physical cores available to this process, CPU model string)"""
    model, cores = "unknown", set()
    try:
        allowed = os.sched_getaffinity(0)
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur and int(cur["processor"]) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                cur = {}
                continue
            k, v = [x.strip() for x in line.split(":", 1)]
            cur[k] = v
            if k == "model name":
                model = v
        if "processor" in cur and int(cur["processor"]) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
    except Exception:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    return n, model


def cpu_baseline(extra=None):
    """SURVEY 8(d) protocol: BASELINE configs[0] exactly -- DeiT-tiny, 1 teacher (dinov2-large), batch 8, fp32, world 1,
    fwd + loss + bwd, 10 warm-up + 50 timed steps, median step time, torch threads = physical cores."""
    import torch
    from oracle import theia_oracle as O
    cores, cpu_model = host_cpu()
    bb, teachers, B = "facebook/deit-tiny-patch16-224", O.TEACHER_SETS["dinov2"], 8
    params = O.synth_params(bb, teachers, 0)
    images = O.synth_images(B, 0)
    targets = O.synth_targets(B, teachers, 1)

    def run(threads, warm, timed):
        torch.set_num_threads(threads)
        for _ in range(warm):
            O.train_step_grads(params, images, targets, bb, teachers)
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            O.train_step_grads(params, images, targets, bb, teachers)
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    # torch's CPU kernels stop scaling around 32 threads for these op sizes and collapse beyond (measured on the 128-core
    # EPYC 9575F host: 2.0 s/step at 128 threads, 0.30 s at 32): the protocol run uses min(physical cores, 32) threads and a
    # 3-step probe at all physical cores is reported beside it; `cores` is the thread count of the reported value
    threads = min(cores, 32)
    med = run(threads, 10, 50)
    out = {"value": round(B / med, 2), "unit": "images/sec", "cores": threads, "kind": "port", "cpu_model": cpu_model,
           "host_physical_cores": cores,
           "sample": "BASELINE configs[0]: DeiT-tiny + dinov2-large, batch 8, fp32, fwd+loss+bwd (no optimizer), 10 warm-up + 50 timed "
                     f"steps, median step {med * 1e3:.1f} ms, torch.set_num_threads({threads})"}
    if cores > threads:
        med_all = run(cores, 1, 3)
        out["value_all_physical_cores"] = round(B / med_all, 2)
        if med_all < med:
            out["value"], out["cores"] = round(B / med_all, 2), cores
    if extra:
        out.update(extra)
    return out


# ------------------------------------------------------------------------------------------------ self-checks
def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def selfcheck_oracle(backbone, precision, dev):
    """bf16 step at batch 2, every GEMM on the ping-pong kernel, against the CPU oracle (fp32).  Returns (report, cpu sample)."""
    import torch
    from oracle import theia_oracle as O  # checker only
    from theia_amd import ops
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    B = 2
    m = RobotVisionFM(backbone=backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                      target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in TEACHERS}, precision=precision)
    params = O.synth_params(backbone, TEACHERS, 0)
    m.load_state_dict(params, strict=True)
    m = m.to(dev)
    images = O.synth_images(B, 0)
    tcpu = O.synth_targets(B, TEACHERS, 1)
    prev, ops.GEMM_TILE_HINT = ops.GEMM_TILE_HINT, 256256
    try:
        losses = m.get_loss(m(images), {t: v.to(dev) for t, v in tcpu.items()})
        main = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
        main.backward()
        torch.cuda.synchronize()
    finally:
        ops.GEMM_TILE_HINT = prev
    cores, _ = host_cpu()
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    _, ref_main, grads, _ = O.train_step_grads(params, images, tcpu, backbone, TEACHERS, "cos_l1")
    dt_cpu = time.perf_counter() - t0
    tol = {"bf16": 2e-2, "fp8": 5e-2}.get(precision, 1e-4)
    rel = abs(float(main) - float(ref_main)) / abs(float(ref_main))
    worst, who = 1.0, ""
    for k, p in m.named_parameters():
        if grads[k].numel() < 4096 or "k_proj" in k:
            continue
        c = _cos(p.grad.detach().float().cpu(), grads[k])
        if c < worst:
            worst, who = c, k
    ok = rel < tol and worst > {"bf16": 0.98, "fp8": 0.9}.get(precision, 0.9999)
    rep = {"ok": bool(ok), "batch": B, "loss_rel_err": float(f"{rel:.3e}"), "worst_grad_cosine": round(worst, 5), "worst_param": who,
           "tile": "256x256 forced"}
    sample = {"oracle_step_same_model": {"value": round(B / dt_cpu, 3), "unit": "images/sec", "cores": cores,
                                         "sample": f"1 fwd+loss+bwd step of {backbone.split('/')[-1]} + 5 teachers at batch {B} (fp32 oracle)"}}
    del m
    torch.cuda.empty_cache()
    return rep, sample


def selfcheck_dispatch(fwd_bwd, engine):
    """the bench-size step: library dispatch vs every GEMM on the 2-stage 128x128 kernel.  The two executions are the same arithmetic
    up to the f32 summation order (the ping-pong kernel starts its accumulators from the bias / residual row): every GEMM output
    agrees to one bf16 ulp in all but ~1e-4 of its elements (selfcheck_gemm_ulp below is the sharp check), and over the 12 layers the
    two steps decorrelate to the bf16 noise floor -- bucket cosines of 0.994-0.999 (tests/test_model_gpu.py::
    test_bf16_bench_dispatch_agrees_with_the_2stage_kernels documents the measurement)."""
    import torch
    from theia_amd import ops
    la = float(fwd_bwd().detach().float())
    torch.cuda.synchronize()
    snap = [b.flat.clone() for b in engine.buckets if b.flat is not None]
    prev, ops.GEMM_TILE_HINT = ops.GEMM_TILE_HINT, 128128
    try:
        lb = float(fwd_bwd().detach().float())
        torch.cuda.synchronize()
    finally:
        ops.GEMM_TILE_HINT = prev
    live = [b for b in engine.buckets if b.flat is not None]
    cur = [b.flat for b in live]
    worst = min(_cos(a, b) for a, b in zip(snap, cur))
    # per TENSOR as well (review item 9: the gate of the model-level tests, at the batch that is benched): every parameter of >= 4096
    # elements, its gradient under the library's dispatch against the same gradient on the 2-stage kernels
    worst_t, worst_name = 1.0, ""
    for b, sa in zip(live, snap):
        for name, off, prm in zip(b.names, b.offsets, b.params):
            n = prm.numel()
            if n < 4096:
                continue
            c = _cos(sa[off:off + n], b.flat[off:off + n])
            if c < worst_t:
                worst_t, worst_name = c, name
    rel = abs(la - lb) / abs(lb)
    # a single wrong tensor inside a large bucket cannot hide behind the bucket cosine: every tensor of >= 4096 elements > 0.98 (the gate
    # of the model-level tests at their small batches; 0.9938 measured at the benched batch), the failing tensor is named
    # fp8 mode: the launches that keep bf16 operands (3x3 kernel, attention-adjacent projections, weight gradients) move to the 2-stage
    # kernels under the hint, and a one-ulp bf16 difference upstream flips e4m3 roundings (3 mantissa bits) downstream: the two executions
    # decorrelate to the e4m3 noise floor instead of the bf16 one (0.969-0.976 measured; the mode's own gate against the oracle is 0.9)
    fp8 = getattr(engine, "precision", "") == "fp8"
    gate_b, gate_t = (0.95, 0.93) if fp8 else (0.99, 0.98)
    return {"ok": bool(rel < 2e-3 and worst > gate_b and worst_t > gate_t), "loss_rel_diff": float(f"{rel:.3e}"), "worst_bucket_cosine": round(worst, 6),
            "worst_tensor_cosine": round(worst_t, 6), "worst_tensor": worst_name, "bucket_gate": gate_b, "tensor_gate": gate_t}


def selfcheck_gemm_ulp(batch, dev):
    """the sharp kernel check at the bench's own GEMM shapes: persistent ping-pong kernel (the library's choice of tile height) vs the
    2-stage 128x128 kernel on the same bf16 operands -- outputs within one bf16 ulp everywhere, different in < 1e-3 of the elements"""
    import math
    import torch
    from theia_amd import ops, _native as N
    torch.manual_seed(1)
    M = batch * 197
    worst_frac, worst_ulp = 0.0, 0.0
    for (Nn, K, kind) in ((768, 768, "resid"), (3072, 768, "gelu"), (768, 3072, "resid"), (2304, 768, "plain"), (3072, 768, "dgelu")):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        bias = torch.randn(Nn, device=dev) * 0.1
        extra = torch.randn(M, Nn, device=dev).to(torch.bfloat16)
        kw = {"resid": dict(resid=extra), "gelu": dict(act=N.ACT_GELU), "plain": {}, "dgelu": dict(act=N.ACT_MUL_DGELU, aux_in=extra)}[kind]
        ya = ops.linear(x, w, None if kind == "dgelu" else bias, **kw).float()
        yb = ops.linear(x, w, None if kind == "dgelu" else bias, tile=128128, **kw).float()
        ulp = torch.maximum(ya.abs(), yb.abs()) * 2.0 ** -7 + 1e-5 * float(yb.abs().max())
        worst_ulp = max(worst_ulp, float(((ya - yb).abs() / ulp).max()))
        worst_frac = max(worst_frac, float((ya != yb).float().mean()))
        del x, w, extra, ya, yb
    torch.cuda.empty_cache()
    return {"ok": bool(worst_ulp <= 1.0 + 1e-6 and worst_frac < 1e-3), "worst_ulp": round(worst_ulp, 3), "differing_fraction": float(f"{worst_frac:.2e}")}


def src_sha(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# the sources of the persistent NT kernel itself (not gemm.hip: host dispatch + the 2-stage kernel -- an edit there voided round 3's summary)
NT_KERNEL_SOURCES = ["theia_amd/csrc/gemm_pp.hip", "theia_amd/csrc/gemm_epi_direct.h", "theia_amd/csrc/gemm_tile.h"]


TRAFFIC_WORKLOAD = ("facebook/deit-base-patch16-224", 128, "bf16")  # what tools/pmc_bench_traffic.sh profiles: the default run


def load_traffic(pfx, workload=TRAFFIC_WORKLOAD, teachers="cddsv"):
    """HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read inside a timed run, so the number
    comes from a committed summary of tools/pmc_bench_traffic.sh -- and only if that summary was taken from the kernel
    sources this run was built from (it records their hash) and on this run's workload (it records that too; summaries of
    rounds 1-4 carry no workload: they are the default one); otherwise null.  The highest round's matching summary wins."""
    want_wl = [workload[0], int(workload[1]), workload[2], teachers]
    default_wl = [TRAFFIC_WORKLOAD[0], TRAFFIC_WORKLOAD[1], TRAFFIC_WORKLOAD[2], "cddsv"]
    pdir = os.path.join(ROOT, "profiles")
    stale = None
    for name in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        if "_bench_pmc_traffic" not in name or not name.endswith(".json"):
            continue
        try:
            tj = json.load(open(os.path.join(pdir, name)))
            if tj.get("_workload", default_wl) != want_wl:
                continue
            if tj.get("_kernel_src_sha") != src_sha(NT_KERNEL_SOURCES):
                stale = stale or name + " (stale: kernel sources changed since it was taken)"
                continue
            want = {"bf16": "gemm_nt_pp_kernel<bf16, *> (all instantiations)", "f32": "gemm_nt_pp_kernel<float, *> (all instantiations)",
                    "fp8": "gemm_nt_pp_kernel<fp8_t, *> (all instantiations)"}[pfx]
            global _KERNEL_TRACE
            _KERNEL_TRACE = dict(tj["_kernel_trace"], source=name) if isinstance(tj.get("_kernel_trace"), dict) and pfx == "bf16" else None
            return (tj[want]["hbm_bytes_per_launch"] if want in tj else None), name
        except Exception:  # a malformed summary must not break the benchmark
            continue
    return None, stale


# rocprofv3 --kernel-trace durations of the dominant kernel recorded beside the matching PMC summary (tools/add_trace_to_traffic.py):
# HIP events on a shared queue include the wait for CUs held by side-stream workgroups, the kernel trace does not
_KERNEL_TRACE = None


# ------------------------------------------------------------------------------------------------ train-step bench
_DIAG = {}  # what an N > 1 run knows about itself so far: printed with the error if a rank fails (review item 7)


def main(argv=None):
    """`_main` with a net under it for N > 1: a rank that fails prints ONE JSON line (rank 0: stdout, others: stderr) with the error, how
    many ranks RCCL connected, the exchange settings, the bucket order and the start-up measurements taken so far -- the driver's scaling
    runs are the only place the RCCL path ever executes, and a bare traceback from one of eight ranks says none of that."""
    try:
        return _main(argv)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        if world > 1:
            line = {"metric": METRIC, "value": None, "unit": "images/sec", "n_gpus": world, "failed_rank": rank,
                    "error": f"{type(e).__name__}: {e}"[:2000], **_DIAG}
            print(json.dumps(line), file=sys.stdout if rank == 0 else sys.stderr, flush=True)
        raise


def _main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_spawn(args, argv)
    TEACHERS[:] = TEACHER_SETS[args.teachers]
    global METRIC
    if (args.backbone, args.teachers) != (BACKBONE, "cddsv"):  # the BASELINE metric string names DeiT-base + 5 teachers
        METRIC = (f"images/sec train-step (fwd+bwd+allreduce) {args.backbone.split('/')[-1]} {len(TEACHERS)}-teacher ({args.teachers})")
    import torch
    import torch.distributed as dist
    if os.environ.get("THEIA_BENCH_DEBUG"):  # dump every thread's stack if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["THEIA_BENCH_DEBUG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test hook (not a product path): THEIA_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 over gloo, so that the N>1 code path
    # of this script (rendezvous, barriers, bucket all-reduces, max-over-ranks timing) can be smoke-tested on a 1-GPU box
    one_device = os.environ.get("THEIA_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            from theia_amd.parallel import configure_rccl_env
            configure_rccl_env()  # THEIA_RCCL_MAX_NCHANNELS -> NCCL_MAX_NCHANNELS
            dist.init_process_group("nccl", device_id=dev)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)  # the first collective: counts the ranks RCCL actually connected
        rccl_ranks = int(one.item())
        log(f"process group up: backend {dist.get_backend()}, {rccl_ranks} ranks answered the first all-reduce")
        _DIAG["rccl_ranks"] = rccl_ranks
    if args.mode == "forward_feature":
        return forward_feature_main(args, rank, world, dev)

    from theia_amd import ops
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.parallel import TheiaDataParallel

    checks = {}
    cpu_extra = None
    if not args.no_selfcheck:
        if rank == 0:
            log("self-check 1: batch-2 step on the ping-pong kernels vs the CPU oracle ...")
            checks["oracle"], cpu_extra = selfcheck_oracle(args.backbone, args.precision, dev)
            log(f"self-check 1: {checks['oracle']}")
        if world > 1:
            dist.barrier()

    torch.manual_seed(0)
    model = RobotVisionFM(backbone=args.backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in TEACHERS},
                          precision=args.precision).to(dev)
    log(f"model built ({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params)")
    ddp = TheiaDataParallel(model)
    if world > 1:
        _DIAG["dp"] = {"backend": ddp.reducer.backend, "exchange": ddp.reducer.exchange, "comm_dtype": ddp.reducer.comm_dtype,
                       "rccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                       "buckets_mb": [[bk.name, round(bk.numel * 4 / 1e6, 1)] for bk in model.engine.buckets]}
    opt = FusedAdamW(ddp, lr=2e-3 * (args.batch * world) / (64 * 8), betas=(0.9, 0.999), weight_decay=0.01)

    b = args.batch
    g = torch.Generator(device="cpu").manual_seed(rank)
    images = torch.randint(0, 256, (b, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    g2 = torch.Generator(device="cpu").manual_seed(1000 + rank)
    targets = {}
    teacher_dtype = torch.float32 if args.teacher_dtype == "fp32" or (args.teacher_dtype == "auto" and args.precision == "fp32") else torch.bfloat16
    for t in TEACHERS:
        C, H, W = get_model_feature_size(t, keep_spatial=True)
        targets[t] = torch.randn(b, H * W, C, generator=g2).to(dev).to(teacher_dtype)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        pred = ddp(images)
        losses = model.get_loss(pred, targets, as_float=False)
        main_loss = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
        main_loss.backward()
        return main_loss

    def step():
        main_loss = fwd_bwd()
        if not args.no_optimizer:
            if opt.capturable:  # (--graph: the optimizer reads its per-step scalars from the device in eager calls too)
                opt.prepare_step()
            opt.step()
        return main_loss

    eager_step = step
    if args.graph:
        if args.no_optimizer:
            raise SystemExit("--graph: with the optimizer (the captured step is the whole step)")
        from theia_amd.train_graph import CapturedTrainStep
        # world > 1: two captured halves with the eager bucket exchange between them (theia_amd/train_graph.py)
        captured = CapturedTrainStep(ddp if world > 1 else model, opt, warmup=2)
        # the synthetic batch lives in the step's static input buffers (what a data pipeline that ingests straight into them does): the
        # eager loop reads its resident batch in place too
        xs, ys = captured.static_inputs(images, targets)
        xs.copy_(images)
        for t_ in targets:
            ys[t_].copy_(targets[t_])
        images, targets = xs, ys

        def step():  # noqa: F811
            return captured(images, targets)["main_loss"]

    log("inputs ready")
    if not args.no_selfcheck:
        checks["dispatch"] = selfcheck_dispatch(fwd_bwd, model.engine)
        log(f"self-check 2: {checks['dispatch']}")
        if args.precision == "bf16" and model.backbone.model.hidden_size == 768:
            checks["gemm_ulp"] = selfcheck_gemm_ulp(b, dev)
            log(f"self-check 3: {checks['gemm_ulp']}")
    ok_flag = torch.tensor([1.0 if all(c["ok"] for c in checks.values()) else 0.0], device=dev)
    if world > 1:
        dist.all_reduce(ok_flag, op=dist.ReduceOp.MIN)
    if float(ok_flag.item()) < 0.5:
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "images/sec", "n_gpus": world, "selfcheck": checks,
                              "error": "self-check failed: the kernels of this run do not reproduce the oracle; no value reported"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(1)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    # settle: a fresh process on a fresh box runs ~15-20 % slow for its first seconds (clock ramp / allocator growth; measured:
    # 70 vs 59 ms/step with 3 warm-up steps, 59.8 with 40).  Keep stepping, untimed, until 3 consecutive steps agree with
    # the previous 3 within 2 % (at most 60 extra steps), then time exactly --steps steps.
    settle_steps, prev3 = 0, None
    while settle_steps < 60:
        ts = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        cur3 = time.perf_counter() - ts
        settle_steps += 3
        stable = prev3 is not None and abs(cur3 - prev3) <= 0.02 * prev3
        flag = torch.tensor([1.0 if stable else 0.0], device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank leaves the loop in the same iteration
        if float(flag.item()) > 0.5:
            break
        prev3 = cur3
    log(f"settled after {settle_steps} extra untimed steps")
    dp_tune = {}
    if world > 1 and not args.no_dp_autotune:
        # untimed: how many CUs to leave to RCCL during the gradient exchange, measured on this job's own step (parallel.py)
        dp_tune = ddp.autotune_reserved_cus(step)
        _DIAG["dp"].update({"reservation_autotune_ms_per_step": dp_tune or None, "cus_left_to_rccl_during_backward": ddp._reserve,
                            "work_conserving_tile_schedule": {"on": bool(ddp.dynamic_schedule), "autotune_ms_per_step": ddp.autotune_dynamic_ms}})
        log(f"CU reservation for the gradient exchange: ms/step per candidate {dp_tune} -> {ddp._reserve} CUs; work-conserving tile "
            f"schedule: {ddp.autotune_dynamic_ms} ms/step -> {'on' if ddp.dynamic_schedule else 'off'}")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(last.detach().float().cpu())
    # host time to enqueue ONE step into empty queues (untimed extra steps): the floor the launch path puts under a step
    host_dt = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        host_dt = min(host_dt, time.perf_counter() - th)
    torch.cuda.synchronize()
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step (host enqueue of one step: {host_dt * 1e3:.2f} ms)")

    roofline = student = None
    if not args.no_roofline:
        pfx = {"bf16": "bf16", "fp8": "fp8"}.get(args.precision, "f32")

        def measure(n_steps):
            """HIP-event durations of every theia_gemm_nt launch over n_steps instrumented steps."""
            ops.GEMM_PROFILE = []
            for _ in range(n_steps):
                eager_step()  # (per-launch HIP events cannot be recorded inside a graph replay)
            torch.cuda.synchronize()
            recs, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            return [(e0.elapsed_time(e1) * 1e-3, fl, var, shp, ab) for (e0, e1, fl, var, shp, ab) in recs]

        def table(label, rr, NP):
            by = {}
            for t_, fl, var, shp, *_ab in rr:
                d = by.setdefault((var,) + shp, [0, 0.0, 0.0])
                d[0] += 1
                d[1] += t_
                d[2] += fl
            log(f"{label}: total {sum(v[1] for v in by.values()) / NP * 1e3:.2f} ms/step, "
                f"{sum(v[2] for v in by.values()) / sum(v[1] for v in by.values()) / 1e12:.1f} TF")
            for k, (cnt, tt_, ff) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                log(f"{label} {k}: {cnt // NP:4d}/step  {tt_ / cnt * 1e6:8.1f} us  {ff / tt_ / 1e12:7.1f} TF  {tt_ / NP * 1e3:7.2f} ms/step")

        NP = 2
        want_table = bool(os.environ.get("THEIA_BENCH_GEMM_TABLE")) and rank == 0
        recs = measure(NP)  # same regime as the timed steps (weight-gradient kernels overlap on the side stream)
        # the persistent ping-pong kernel (gemm_nt_pp_kernel) runs 256- and 320-row tiles: one kernel, two tile heights
        fam = lambda v: "pingpong" if v in ("256x256", "320x256") else v
        recs = [(t_, f_, fam(v_), s_, ab_) for t_, f_, v_, s_, ab_ in recs]
        tot_by_var = {}
        for t_, _f, v_, _s, _ab in recs:
            tot_by_var[v_] = tot_by_var.get(v_, 0.0) + t_
        dom_var = max(tot_by_var, key=tot_by_var.get)  # the tile variant with the largest share of the step
        dom = [(t_, f_) for t_, f_, v_, _s, _ab in recs if v_ == dom_var]
        alg_bytes = [ab_ for _t, _f, v_, _s, ab_ in recs if v_ == dom_var]
        if want_table:
            table("gemm_nt", recs, NP)
        tsum, fsum = sum(t for t, _ in dom), sum(f for _, f in dom)
        achieved = fsum / tsum / 1e12
        # the same launches with the side stream switched off: the kernel alone on the chip
        sq = getattr(model.engine, "_sideq", None)
        if sq is not None and sq.enabled:
            sq.join()
            torch.cuda.synchronize()
            sq.enabled = False
            if want_table:
                ops.WGRAD_PROFILE = []
            iso_recs = measure(NP)
            iso = [(t_, f_) for t_, f_, v_, _s, _ab in iso_recs if fam(v_) == dom_var]
            if ops.WGRAD_PROFILE is not None:  # isolated per-shape tables (tuning aid)
                wrecs, ops.WGRAD_PROFILE = ops.WGRAD_PROFILE, None
                table("gemm_nt(isolated)", iso_recs, NP)
                table("gemm_wgrad(isolated)", [(e0.elapsed_time(e1) * 1e-3, fl, var, shp, 0) for (e0, e1, fl, var, shp) in wrecs], NP)
            sq.enabled = True
            iso_tf = sum(f for _, f in iso) / sum(t for t, _ in iso) / 1e12
        else:
            iso_tf = achieved
        kname = (f"gemm_nt_pp_kernel<{pfx}> (theia_gemm_nt: persistent ping-pong kernel, 256x256 / 320x256 tiles)" if dom_var == "pingpong"
                 else f"gemm_nt_kernel<{pfx},{dom_var.replace('x', ',')}> (theia_gemm_nt)")
        traffic, traffic_src = load_traffic(pfx, (args.backbone, b, args.precision), args.teachers) if dom_var == "pingpong" else (None, None)
        # Which roofline bounds the dominant kernel: its arithmetic intensity (algorithmic flop per algorithmic byte, both per launch,
        # mean over the step's launches) against the ridge of the part (2.5 PFLOP/s / 8 TB/s = 312 flop/B).  DeiT-base: ~565 flop/B ->
        # MFMA; DeiT-tiny (K = 192: a [50432, 768] x [768, 192] launch moves 175 MB for 15 GFLOP, 85 flop/B) -> HBM, and is reported
        # against that bound: achieved = algorithmic bytes / launch duration.
        alg_mean = sum(alg_bytes) / max(1, len(alg_bytes))
        intensity = (fsum / len(dom)) / max(1.0, alg_mean)
        if intensity < MFMA_BF16_PEAK / HBM_PEAK:
            iso_pairs = iso if (sq is not None and sq.enabled) else dom
            gbs = sum(alg_bytes) / tsum / 1e9
            roofline = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": round(gbs * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "traffic_algorithmic": round(alg_mean), "arithmetic_intensity_flop_per_byte": round(intensity, 1),
                        "launches_per_step": len(dom) // NP, "avg_launch_us": round(tsum / len(dom) * 1e6, 1),
                        "flops_per_launch": round(fsum / len(dom)),
                        "achieved_isolated": round(sum(alg_bytes) / sum(t for t, _ in iso_pairs) / 1e9, 1),
                        "frac_isolated": round(sum(alg_bytes) / sum(t for t, _ in iso_pairs) / HBM_PEAK, 4),
                        "frac_of_copy_rate_6300": round(sum(alg_bytes) / sum(t for t, _ in iso_pairs) / 6.3e12, 4),
                        "achieved_tflops": round(achieved, 1),
                        "note": "HBM-bound kernel (intensity below the 312 flop/B ridge): achieved = algorithmic bytes per launch / launch "
                                "duration; *_isolated = same launches with the side-stream overlap off; 6.3 TB/s = what a copy reaches"}
        else:
          roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK / 1e12,
                    "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_BF16_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                    # minimum HBM bytes per launch (mean over the same launches): operands read once + outputs written once + the
                    # epilogue's row inputs (ops.gemm_nt_algorithmic_bytes); `traffic` / this = re-read factor
                    "traffic_algorithmic": round(sum(alg_bytes) / max(1, len(alg_bytes))),
                    "launches_per_step": len(dom) // NP, "avg_launch_us": round(tsum / len(dom) * 1e6, 1),
                    "flops_per_launch": round(fsum / len(dom)),
                    "achieved_isolated": round(iso_tf, 1), "frac_isolated": round(iso_tf * 1e12 / MFMA_BF16_PEAK, 4),
                    # context, not the judged fraction: the chip is power-limited under MFMA streams of random bf16 operands -- an MFMA-only
                    # stream with no data movement reaches 1513 TFLOP/s at a 1.44 GHz shader clock (tools/experiments/w4_probe_not_kept.patch,
                    # profiles/r05_one_wave_per_simd_kernel_not_kept.txt); `peak` stays the 2.5 PFLOP/s of the microarchitecture guide
                    "power_limited_peak_r05_constant": 1513.0, "frac_isolated_of_power_limited_peak": round(iso_tf / 1513.0, 4),
                    "arithmetic_intensity_flop_per_byte": round(intensity, 1),
                    "note": "achieved/avg_launch_us are measured in the regime of the timed steps (weight-gradient kernels run "
                            "concurrently on a side stream and share the CUs); *_isolated = same launches with that overlap off"}
        if _KERNEL_TRACE is not None and roofline["bound"] == "mfma":
            # the rocprof-comparable figures: this run's algorithmic flop per launch / the committed kernel-trace durations (same kernel
            # sources, same workload; the averages of profiles/*_kernel_stats{,_serial}.csv)
            kt = {"source": _KERNEL_TRACE["source"]}
            for k_ in ("regime", "serial"):
                r_ = _KERNEL_TRACE.get(k_)
                if r_:
                    kt[k_] = {"avg_launch_us": r_["avg_us"], "calls": r_["calls"],
                              "frac": round(roofline["flops_per_launch"] / (r_["avg_us"] * 1e-6) / MFMA_BF16_PEAK, 4)}
            roofline["rocprof_kernel_trace"] = kt
        # the student alone: backbone forward + backward (weight gradients on the side stream as in the full step)
        if args.backbone == BACKBONE:
            dz = torch.randn(b, 197, 768, device=dev).to(model.engine.dtype) * 1e-3

            def student_step():
                opt.zero_grad(set_to_none=True)
                z = model.backbone(images)
                z.backward(dz)

            for _ in range(2):
                student_step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            NS = 5
            e0.record()
            for _ in range(NS):
                student_step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / NS
            tf = FLOPS_PER_IMAGE_STUDENT * b / (ms * 1e-3) / 1e12
            student = {"bound": "mfma", "what": "DeiT-base student alone: patch-embed + 12 layers + final LN, forward + backward "
                                               "(data and weight gradients), per-GPU batch %d" % b,
                       "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(tf * 1e12 / MFMA_BF16_PEAK, 4),
                       "ms": round(ms, 3), "flops_per_image": FLOPS_PER_IMAGE_STUDENT}

    if rank == 0:
        imgs = world * b * args.steps
        value = imgs / dt
        out = {
            "metric": METRIC,
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp8": "fp8 e4m3 operands (forward + data-gradient GEMMs of the linears and transposed convolutions; the one-image-per-tile 3x3 "
                             "convolutions, the attention-adjacent projections and every weight gradient keep bf16 operands), f32 accumulate, bf16 out"}.get(args.precision, "f32"),
            "data": "synthetic",
            "config": {"workload": f"{args.backbone.split('/')[-1]} student + {len(TEACHERS)} teacher{'s' if len(TEACHERS) > 1 else ''} ({args.teachers}), per-GPU batch {b}, "
                                   f"teacher features resident as {'bf16' if teacher_dtype == torch.bfloat16 else 'fp32'}, "
                                   f"loss 0.9*cos+0.1*smoothL1, step = fwd+loss+bwd+grad all-reduce" +
                                   ("" if args.no_optimizer else "+fused AdamW") + ((", one hipGraph replay per step" if world == 1 else ", two hipGraph replays per step around the eager gradient exchange") if args.graph else ""),
                       "global_batch": b * world, "parallelism": f"dp{world}", "final_loss": round(loss_val, 5)},
            "settle_steps": settle_steps,
            "host_enqueue_ms_per_step": round(host_dt * 1e3, 3),
            "rccl_ranks": rccl_ranks,
            "dp": None if world == 1 else {"backend": ddp.reducer.backend, "exchange": ddp.reducer.exchange, "comm_dtype": ddp.reducer.comm_dtype,
                                           "rccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                                           "cus_left_to_rccl_during_backward": ddp._reserve,
                                           "reservation_autotune_ms_per_step": dp_tune or None,
                                           "work_conserving_tile_schedule": {"on": bool(ddp.dynamic_schedule), "autotune_ms_per_step": ddp.autotune_dynamic_ms},
                                           # exchange order = backward-completion order; MB of fp32 per bucket
                                           "buckets_mb": [[bk.name, round(bk.numel * 4 / 1e6, 1)] for bk in model.engine.buckets]},
            "selfcheck": checks if checks else "skipped (--no-selfcheck): unchecked run",
            "model_flops_utilization": round(value / world * FLOPS_PER_IMAGE[(args.backbone, args.teachers)] / MFMA_BF16_PEAK, 4)
            if FLOPS_PER_IMAGE.get((args.backbone, args.teachers)) else None,
        }
        if roofline is not None:
            out["roofline"] = roofline
        if student is not None:
            out["student_roofline"] = student
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores, BASELINE configs[0] protocol) ...")
            out["cpu_baseline"] = cpu_baseline(cpu_extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


FF_METRIC = "forward_feature images/sec DeiT-base batch 4096 streamed hipGraph"


def forward_feature_main(args, rank, world, dev):
    """BASELINE configs[4]: forward_feature() inference throughput -- DeiT-base student, 4096 uint8 images resident in HBM,
    streamed in chunks through ONE hipGraph capture of the chunk's forward (theia_amd/streaming.py); a step = one pass over
    the 4096 images, features written to a resident [4096, 196, 768] f32 tensor.  N > 1: independent replicas (no collective)."""
    import torch
    import torch.distributed as dist
    from theia_amd import ops
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.streaming import StreamedForwardFeature

    torch.manual_seed(0)
    model = RobotVisionFM(backbone=args.backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes=None, precision=args.precision).to(dev).eval()
    D = model.backbone.model.hidden_size
    B, chunk = args.stream_batch, args.chunk
    checks = {}
    cpu = None
    if not args.no_selfcheck and rank == 0:
        from oracle import theia_oracle as O  # checker only
        params = {k: v for k, v in O.synth_params(args.backbone, [], 0).items()}
        model.load_state_dict(params, strict=True)
        small = O.synth_images(8, 0)
        cores, cpu_model = host_cpu()
        torch.set_num_threads(cores)
        with torch.no_grad():
            O.forward_feature(params, small[:2], args.backbone)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                ref = O.forward_feature(params, small, args.backbone)
                ts.append(time.perf_counter() - t0)
            eager = model.forward_feature(small)
        got = StreamedForwardFeature(model, chunk=3)(small.to(dev))
        err = float((eager.float().cpu() - ref).abs().max() / ref.abs().max())
        checks = {"ok": bool(err < (2e-2 if args.precision == "bf16" else 1e-4) and torch.equal(got, eager)),
                  "max_err_vs_oracle_rel": float(f"{err:.3e}"), "graph_replay_equals_eager": bool(torch.equal(got, eager)), "batch": 8}
        cpu = {"value": round(8 / statistics.median(ts), 2), "unit": "images/sec", "cores": cores, "kind": "port", "cpu_model": cpu_model,
               "sample": f"oracle forward_feature of {args.backbone.split('/')[-1]} (fp32 torch CPU), batch 8, median of 5 passes"}
        log(f"self-check: {checks}")
    ok_flag = torch.tensor([1.0 if checks.get("ok", True) else 0.0], device=dev)
    if world > 1:
        dist.all_reduce(ok_flag, op=dist.ReduceOp.MIN)
    if float(ok_flag.item()) < 0.5:
        if rank == 0:
            print(json.dumps({"metric": FF_METRIC, "value": None, "unit": "images/sec", "n_gpus": world, "selfcheck": checks,
                              "error": "self-check failed: no value reported"}), flush=True)
        raise SystemExit(1)

    g = torch.Generator(device="cpu").manual_seed(rank)
    images = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    out = torch.empty(B, 196, D, dtype=torch.float32, device=dev)
    sff = StreamedForwardFeature(model, chunk=chunk)

    def step():
        sff(images, out=out)

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # host time of one streamed pass (what replaces ~110 launches per chunk) and the eager path beside it
    torch.cuda.synchronize()
    th = time.perf_counter()
    step()
    host_dt = time.perf_counter() - th
    torch.cuda.synchronize()
    with torch.no_grad():
        model.forward_feature(images[:chunk])
        torch.cuda.synchronize()
        te = time.perf_counter()
        for i in range(0, B, chunk):
            out[i:i + chunk] = model.forward_feature(images[i:i + chunk])
        torch.cuda.synchronize()
        eager_dt = time.perf_counter() - te
    roofline = None
    if not args.no_roofline:
        # per-launch HIP events cannot be placed inside a graph replay: the same chunk forward, eager, instrumented
        ops.GEMM_PROFILE = []
        with torch.no_grad():
            for _ in range(2):
                model.forward_feature(images[:chunk])
        torch.cuda.synchronize()
        recs, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        rr = [(e0.elapsed_time(e1) * 1e-3, fl, var) for (e0, e1, fl, var, _s, _ab) in recs]
        tot = {}
        for t_, _f, v_ in rr:
            tot[v_] = tot.get(v_, 0.0) + t_
        dom_var = max(tot, key=tot.get)
        tsum = sum(t for t, _f, v in rr if v == dom_var)
        fsum = sum(f for _t, f, v in rr if v == dom_var)
        n = sum(1 for _t, _f, v in rr if v == dom_var)
        pfx = "bf16" if args.precision == "bf16" else "f32"
        roofline = {"bound": "mfma", "kernel": f"gemm_nt tile {dom_var} <{pfx}> (theia_gemm_nt)", "achieved": round(fsum / tsum / 1e12, 1),
                    "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(fsum / tsum / MFMA_BF16_PEAK, 4), "traffic": None,
                    "launches_per_chunk": n // 2, "avg_launch_us": round(tsum / n * 1e6, 1), "flops_per_launch": round(fsum / n),
                    "note": "measured on an eager pass of the same chunk (events cannot be recorded inside a graph replay)"}
    if rank == 0:
        value = world * B * args.steps / dt
        res = {"metric": FF_METRIC, "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": f"forward_feature {args.backbone.split('/')[-1]}, {B} uint8 224x224x3 images resident in HBM, streamed in "
                                      f"chunks of {chunk} through one hipGraph capture, features f32 [B,196,{D}]",
                          "global_batch": B * world, "parallelism": f"replicas x{world}"},
               "graph_replays_per_step": (B + chunk - 1) // chunk, "host_ms_per_step": round(host_dt * 1e3, 3),
               "eager_images_per_sec": round(B / eager_dt, 1), "selfcheck": checks if checks else "skipped",
               "model_flops_utilization": round(value / world * FLOPS_PER_IMAGE_FWD / MFMA_BF16_PEAK, 4) if args.backbone == BACKBONE else None}
        if roofline is not None:
            res["roofline"] = roofline
        if cpu is not None and world == 1:
            res["cpu_baseline"] = cpu
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
