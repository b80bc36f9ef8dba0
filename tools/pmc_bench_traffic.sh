#!/bin/bash
# HBM-side traffic of every kernel of the default bench step: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
# share a pass; kernel-trace only), then tools/pmc_summary.py.  usage: [PMC_BENCH_ARGS="--backbone ... --batch ..."] tools/pmc_bench_traffic.sh <outdir>
# (PMC_BENCH_ARGS: another workload than the default one; the summary records it as "_workload" and bench.py matches on it)
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o pmc -- \
      python $REPO/bench.py ${PMC_BENCH_ARGS:-} --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-selfcheck > $OUT/$c.log 2>&1
done
python $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
python - "$OUT/summary.txt" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
out = {}
agg = {}  # the persistent ping-pong kernel runs as several instantiations (tile height, tap gather, statistics): per-launch mean over all
for block in re.split(r"\n(?=\S)", txt):
    lines = block.strip().split("\n")
    vals, cnts = {}, {}
    for l in lines[1:]:
        m = re.match(r"\s+(\w+)\s+([\d.]+)\s+\(x(\d+)\)", l)
        if m:
            vals[m.group(1)], cnts[m.group(1)] = float(m.group(2)), int(m.group(3))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide streaming reads at half their bytes on gfx950 -> x2; KB units
        name = lines[0].strip()
        out[name] = {"fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                     "hbm_bytes_per_launch": round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)}
        m = re.match(r"(gemm_nt_pp_kernel<(bf16|float|fp8_t))", name)
        if m:
            a = agg.setdefault(m.group(1) + ", *> (all instantiations)", {"f": 0.0, "fn": 0, "w": 0.0, "wn": 0})
            a["f"] += vals["FETCH_SIZE"] * cnts["FETCH_SIZE"]; a["fn"] += cnts["FETCH_SIZE"]
            a["w"] += vals["WRITE_SIZE"] * cnts["WRITE_SIZE"]; a["wn"] += cnts["WRITE_SIZE"]
for name, a in agg.items():
    f, w = a["f"] / a["fn"], a["w"] / a["wn"]
    out[name] = {"fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1), "hbm_bytes_per_launch": round((2.0 * f + w) * 1024)}
# hash of the NT kernel sources the counters were taken from: bench.py only reports `roofline.traffic` from a summary whose
# hash matches the sources it runs (otherwise the number would silently go stale)
import hashlib, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
h = hashlib.sha256()
for f in ("theia_amd/csrc/gemm_pp.hip", "theia_amd/csrc/gemm_epi_direct.h", "theia_amd/csrc/gemm_tile.h"):  # = bench.py NT_KERNEL_SOURCES
    h.update(open(os.path.join(root, f), "rb").read())
out["_kernel_src_sha"] = h.hexdigest()[:16]
# the workload the counters belong to (backbone, per-GPU batch, precision, teacher set): parsed from PMC_BENCH_ARGS with bench.py's defaults
import shlex
a = shlex.split(os.environ.get("PMC_BENCH_ARGS", ""))
def opt(flag, default):
    return a[a.index(flag) + 1] if flag in a else default
out["_workload"] = [opt("--backbone", "facebook/deit-base-patch16-224"), int(opt("--batch", "128")), opt("--precision", "bf16"), opt("--teachers", "cddsv")]
json.dump(out, open(sys.argv[1].replace("summary.txt", "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "gemm" in k}, indent=1))
PY
