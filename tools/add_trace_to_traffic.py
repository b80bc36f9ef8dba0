#!/usr/bin/env python
"""Adds the kernel-trace durations of the persistent NT kernel (all instantiations: calls, mean duration) to a PMC traffic summary, so
that bench.py can print the rocprof-comparable roofline fraction beside its HIP-event one (same source hash, same workload):
    python tools/add_trace_to_traffic.py <traffic.json> <kernel_stats.csv (regime of the timed steps)> <kernel_stats_serial.csv>"""
import csv
import json
import sys


def nt_mean(path, prefix="gemm_nt_pp_kernel<bf16"):
    calls, total = 0, 0.0
    for row in csv.DictReader(open(path)):
        if row["kernel"].startswith(prefix):
            calls += int(row["calls"])
            total += float(row["total_ms"])
    return {"calls": calls, "avg_us": round(total / calls * 1e3, 2)} if calls else None


tj = json.load(open(sys.argv[1]))
tj["_kernel_trace"] = {"regime": nt_mean(sys.argv[2]), "serial": nt_mean(sys.argv[3]),
                       "what": "rocprofv3 --kernel-trace: mean duration of gemm_nt_pp_kernel<bf16, *> launches; regime = weight-gradient "
                               "kernels overlapping on the side stream (the timed steps), serial = THEIA_SIDE_STREAM=0"}
json.dump(tj, open(sys.argv[1], "w"), indent=1)
print(tj["_kernel_trace"])
