#!/usr/bin/env python
"""bench.py -- images/sec of one Theia distillation train step on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (N > 1, one rank per GPU over RCCL)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): DeiT-base-patch16-224 student, 5
teachers (cddsv: ViT-H, DINOv2-L, CLIP-L, SAM-H, Depth-Anything-L), bf16 MFMA operands / f32 accumulate / f32 master
weights, per-GPU batch 128 (global 1024 at 8 GPUs -> weak scaling), loss 0.9*cos + 0.1*smooth-L1.
One step = forward + loss + backward + gradient all-reduce (RCCL, overlapped) + fused AdamW update, on synthetic
uint8 224x224x3 images and random f32 teacher features already resident in HBM.  Random-init weights.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel = the theia_gemm_nt tile variant with the largest share of the step (MFMA-bound): algorithmic FLOPs of its launches /
                their HIP-event-measured duration (events on the launch stream, taken in extra instrumented steps
                right after the timed region so the timed steps stay unperturbed)
  cpu_baseline  the CPU oracle (oracle/theia_oracle.py, torch fp32 on the host cores) on a bounded sample of the
                same workload (same model / teachers / loss, batch 2, 2 steps, <= 32 threads), rank 0, N == 1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

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
FLOPS_PER_IMAGE_FWD_BWD = 272.169e9  # BASELINE.md sec. 3 (2xMAC, fwd+bwd, base + cddsv)
MFMA_BF16_PEAK = 2.5e15              # dense, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--backbone", default=BACKBONE)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="exclude the AdamW update from the step")
    return ap.parse_args()


def cpu_baseline(backbone, teachers):
    """Oracle (CPU restatement) timed on the host cores: bounded sample of the same workload."""
    from oracle import theia_oracle as O
    # 32 threads: torch's CPU kernels stop scaling (and thrash on 256-core hosts) beyond that for these op sizes
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    B = 2
    params = O.synth_params(backbone, teachers, 0)
    images = O.synth_images(B, 0)
    targets = O.synth_targets(B, teachers, 1)
    O.train_step_grads(params, images[:1], {t: v[:1] for t, v in targets.items()}, backbone, teachers)  # warm-up
    n = 2
    t0 = time.perf_counter()
    for _ in range(n):
        O.train_step_grads(params, images, targets, backbone, teachers)
    dt = time.perf_counter() - t0
    return {"value": round(n * B / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} fwd+loss+bwd steps of {backbone.split('/')[-1]} + 5 teachers at batch {B} (fp32 torch CPU oracle, no optimizer)"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if os.environ.get("THEIA_BENCH_DEBUG"):  # dump every thread's stack if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["THEIA_BENCH_DEBUG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (see the docstring)")
    # test hook (not a product path): THEIA_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 over gloo, so that the N>1 code path
    # of this script (rendezvous, barriers, bucket all-reduces, max-over-ranks timing) can be smoke-tested on a 1-GPU box
    # (gloo with device tensors and two processes per GPU dead-locks sporadically on this stack: use THEIA_BENCH_DEBUG=<s>)
    one_device = os.environ.get("THEIA_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from theia_amd import ops
    from theia_amd.foundation_models.common import get_model_feature_size
    from theia_amd.models.rvfm import RobotVisionFM
    from theia_amd.optimizers import FusedAdamW
    from theia_amd.parallel import TheiaDataParallel

    torch.manual_seed(0)
    model = RobotVisionFM(backbone=args.backbone, translator="lconv", translator_kwargs={"hidden_size_factor": 1.0},
                          target_feature_sizes={t: get_model_feature_size(t, keep_spatial=True) for t in TEACHERS},
                          precision=args.precision).to(dev)
    log(f"model built ({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params)")
    ddp = TheiaDataParallel(model)
    opt = FusedAdamW(ddp, lr=2e-3 * (args.batch * world) / (64 * 8), betas=(0.9, 0.999), weight_decay=0.01)

    b = args.batch
    g = torch.Generator(device="cpu").manual_seed(rank)
    images = torch.randint(0, 256, (b, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    g2 = torch.Generator(device="cpu").manual_seed(1000 + rank)
    targets = {}
    for t in TEACHERS:
        C, H, W = get_model_feature_size(t, keep_spatial=True)
        targets[t] = torch.randn(b, H * W, C, generator=g2).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        pred = ddp(images)
        losses = model.get_loss(pred, targets, as_float=False)
        main_loss = 0.9 * losses["cos_loss"] + 0.1 * losses["l1_loss"]
        main_loss.backward()
        if not args.no_optimizer:
            opt.step()
        return main_loss

    log("inputs ready")
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

    roofline = None
    if not args.no_roofline:
        pfx = "bf16" if args.precision == "bf16" else "f32"

        def measure(n_steps):
            """HIP-event durations of every theia_gemm_nt launch over n_steps instrumented steps."""
            ops.GEMM_PROFILE = []
            for _ in range(n_steps):
                step()
            torch.cuda.synchronize()
            recs, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            return [(e0.elapsed_time(e1) * 1e-3, fl, var, shp) for (e0, e1, fl, var, shp) in recs]

        NP = 2
        recs = measure(NP)  # same regime as the timed steps (weight-gradient kernels overlap on the side stream)
        tot_by_var = {}
        for t_, _f, v_, _s in recs:
            tot_by_var[v_] = tot_by_var.get(v_, 0.0) + t_
        dom_var = max(tot_by_var, key=tot_by_var.get)  # the tile variant with the largest share of the step
        dom = [(t_, f_) for t_, f_, v_, _s in recs if v_ == dom_var]
        if os.environ.get("THEIA_BENCH_GEMM_TABLE") and rank == 0:  # per-shape table on stderr (tuning aid)
            by = {}
            for t_, fl, var, shp in recs:
                d = by.setdefault((var,) + shp, [0, 0.0, 0.0])
                d[0] += 1
                d[1] += t_
                d[2] += fl
            for k, (cnt, tt, ff) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                log(f"gemm_nt {k}: {cnt // NP:4d}/step  {tt / cnt * 1e6:8.1f} us  {ff / tt / 1e12:7.1f} TF  {tt / NP * 1e3:7.2f} ms/step")
        tsum, fsum = sum(t for t, _ in dom), sum(f for _, f in dom)
        achieved = fsum / tsum / 1e12
        # the same launches with the side stream switched off: the kernel alone on the chip
        sq = getattr(model.engine, "_sideq", None)
        if sq is not None and sq.enabled:
            sq.join()
            torch.cuda.synchronize()
            sq.enabled = False
            if os.environ.get("THEIA_BENCH_GEMM_TABLE") and rank == 0:
                ops.WGRAD_PROFILE = []
            iso_recs = measure(NP)
            iso = [(t_, f_) for t_, f_, v_, _s in iso_recs if v_ == dom_var]
            if ops.WGRAD_PROFILE is not None:  # isolated per-shape tables (tuning aid)
                wrecs, ops.WGRAD_PROFILE = ops.WGRAD_PROFILE, None
                for label, rr in (("gemm_nt(isolated)", iso_recs),
                                  ("gemm_wgrad(isolated)", [(e0.elapsed_time(e1) * 1e-3, fl, var, shp) for (e0, e1, fl, var, shp) in wrecs])):
                    by = {}
                    for t_, fl, var, shp in rr:
                        d = by.setdefault((var,) + shp, [0, 0.0, 0.0])
                        d[0] += 1
                        d[1] += t_
                        d[2] += fl
                    log(f"{label}: total {sum(v[1] for v in by.values()) / NP * 1e3:.2f} ms/step, {sum(v[2] for v in by.values()) / sum(v[1] for v in by.values()) / 1e12:.1f} TF")
                    for k, (cnt, tt, ff) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                        log(f"{label} {k}: {cnt // NP:4d}/step  {tt / cnt * 1e6:8.1f} us  {ff / tt / 1e12:7.1f} TF  {tt / NP * 1e3:7.2f} ms/step")
            sq.enabled = True
            iso_tf = sum(f for _, f in iso) / sum(t for t, _ in iso) / 1e12
        else:
            iso_tf = achieved
        kname = (f"gemm_nt_pp_kernel<{pfx}> (theia_gemm_nt, 256x256 ping-pong tile)" if dom_var == "256x256"
                 else f"gemm_nt_kernel<{pfx},{dom_var.replace('x', ',')}> (theia_gemm_nt)")
        # HBM-side bytes per launch of this kernel: PMC numbers cannot be collected inside a timed run, so they come from the
        # committed summary of tools/pmc_bench_traffic.sh (FETCH_SIZE x2 per the guide's gfx950 correction + WRITE_SIZE,
        # mean over the launches of the same bench step); null when the file is absent
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_bench_pmc_traffic.json")
        if dom_var == "256x256" and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = [k for k in tj if k.startswith("gemm_nt_pp_kernel") and (("bf16" in k) == (pfx == "bf16"))]
                traffic = tj[key[0]]["hbm_bytes_per_launch"] if key else None
            except Exception:  # a malformed summary must not break the benchmark
                traffic = None
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK / 1e12,
                    "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_BF16_PEAK, 4), "traffic": traffic,
                    "launches_per_step": len(dom) // NP, "avg_launch_us": round(tsum / len(dom) * 1e6, 1),
                    "flops_per_launch": round(fsum / len(dom)),
                    "achieved_isolated": round(iso_tf, 1), "frac_isolated": round(iso_tf * 1e12 / MFMA_BF16_PEAK, 4),
                    "note": "achieved/avg_launch_us are measured in the regime of the timed steps (weight-gradient kernels run "
                            "concurrently on a side stream and share the CUs); *_isolated = same launches with that overlap off"}

    if rank == 0:
        imgs = world * b * args.steps
        value = imgs / dt
        out = {
            "metric": "images/sec train-step (fwd+bwd+allreduce) DeiT-base 5-teacher",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.backbone.split('/')[-1]} student + 5 teachers (cddsv), per-GPU batch {b}, "
                                   f"loss 0.9*cos+0.1*smoothL1, step = fwd+loss+bwd+grad all-reduce" +
                                   ("" if args.no_optimizer else "+fused AdamW"),
                       "global_batch": b * world, "parallelism": f"dp{world}", "final_loss": round(loss_val, 5)},
            "settle_steps": settle_steps,
            "host_enqueue_ms_per_step": round(host_dt * 1e3, 3),
            "model_flops_utilization": round(value / world * FLOPS_PER_IMAGE_FWD_BWD / MFMA_BF16_PEAK, 4)
            if args.backbone == BACKBONE else None,
        }
        if roofline is not None:
            out["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline(args.backbone, TEACHERS)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
