#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel, mean counter value per dispatch."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("unsigned short", "bf16").replace("void ", "")
    return n[:70]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", r.get("kernel_name", "?")))
            c = r.get("Counter_Name", r.get("counter_name"))
            v = float(r.get("Counter_Value", r.get("counter_value", 0)))
            a = acc[k][c]
            a[0] += v
            a[1] += 1
    for k, cs in acc.items():
        if not any(s in k for s in ("gemm", "attn", "chw", "colsum", "ln_row", "loss")):
            continue
        print(k)
        for c, (s, n) in sorted(cs.items()):
            print(f"    {c:28s} {s / n:16.1f}  (x{n})")


if __name__ == "__main__":
    main()
