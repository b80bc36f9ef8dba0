#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration / share.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--csv out.csv] [--top 40]
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("unsigned short", "bf16").replace("void ", "")
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
    for n, cnt, tot, avg, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{cnt},{tot / 1e6:.3f},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * tot / total:.2f}")
    if a.csv:
        open(a.csv, "w").write("\n".join(lines) + "\n")
    print(f"total kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    for ln in lines[: a.top + 1]:
        print(ln)


if __name__ == "__main__":
    main()
