#!/usr/bin/env python
"""Per-kernel HBM-side rate of a bench step: joins the PMC traffic summary (tools/pmc_bench_traffic.sh: bytes per launch) with a serial
kernel trace (tools/rocpd_stats.py --csv: calls, total_ms) -- GB per step, ms per step, TB/s, fraction of the 6.3 TB/s a copy reaches
(MI355X_MICROARCH.md) and of the 8 TB/s peak.  usage: tools/hbm_fractions.py <traffic.json> <kernel_stats_serial.csv>"""
import csv
import json
import re
import sys


def norm(k):
    return re.sub(r"\s+", "", k.replace("unsigned short", "bf16"))


def main():
    tj = json.load(open(sys.argv[1]))
    rows = list(csv.DictReader(open(sys.argv[2])))
    steps = next(int(r["calls"]) for r in rows if r["kernel"].startswith("cast_batch"))
    calls = {norm(r["kernel"]): (int(r["calls"]) / steps, float(r["total_ms"]) / steps, r["kernel"]) for r in rows}
    out, tot_b, tot_ms = [], 0.0, 0.0
    for k, v in tj.items():
        if not isinstance(v, dict) or "all instantiations" in k:
            continue
        nk = norm(k)
        m = calls.get(nk) or next((c for n, c in calls.items() if n.startswith(nk[:48])), None)
        if m is None:
            continue
        n, ms, name = m
        b = v["hbm_bytes_per_launch"] * n
        tot_b += b
        tot_ms += ms
        out.append((b, n, ms, name))
    out.sort(reverse=True)
    print(f"workload {tj.get('_workload', 'default')}; {steps} steps in the trace; HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE")
    print(f"{'GB/step':>8} {'launches':>9} {'ms/step':>8} {'TB/s':>6} {'of 6.3':>7} {'of 8.0':>7}  kernel")
    for b, n, ms, name in out:
        r = b / ms / 1e9 if ms > 0 else 0.0
        print(f"{b / 1e9:8.2f} {n:9.1f} {ms:8.3f} {r:6.2f} {r / 6.3:7.2f} {r / 8.0:7.2f}  {name[:90]}")
    print(f"{tot_b / 1e9:8.2f} {'':9} {tot_ms:8.3f}  total of the matched kernels; at 6.3 TB/s: {tot_b / 6.3e12 * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
