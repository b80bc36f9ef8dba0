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
    # GPU busy fraction (union of kernel intervals / span) over the last 60% of the trace (steady-state steps)
    iv = c.execute("select start, end from kernels order by start").fetchall()
    if iv:
        t_lo = iv[0][0] + 0.4 * (iv[-1][1] - iv[0][0])
        iv = [(s_, e_) for s_, e_ in iv if s_ >= t_lo]
        busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
        for s_, e_ in iv[1:]:
            if s_ > cur_e:
                busy += cur_e - cur_s
                gaps.append(s_ - cur_e)
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += cur_e - cur_s
        span = iv[-1][1] - iv[0][0]
        gaps.sort()
        print(f"steady-state window {span / 1e6:.1f} ms: GPU busy {100.0 * busy / span:.1f}% ; {len(gaps)} idle gaps, "
              f"total {sum(gaps) / 1e6:.2f} ms, median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, "
              f"{sum(1 for g_ in gaps if g_ > 20000)} gaps > 20 us totalling {sum(g_ for g_ in gaps if g_ > 20000) / 1e6:.2f} ms")
    for ln in lines[: a.top + 1]:
        print(ln)


if __name__ == "__main__":
    main()
