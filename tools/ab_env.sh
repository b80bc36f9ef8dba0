#!/bin/bash
# whole-step A/B of ENVIRONMENT settings on one box, arms alternating: tools/ab_env.sh <outdir> <rounds> "NAME:VAR=val VAR2=val" ...
out=$1; rounds=$2; shift 2
mkdir -p $out
for r in $(seq 1 $rounds); do
  for arm in "$@"; do
    name=${arm%%:*}; envs=${arm#*:}
    env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-selfcheck --no-roofline $AB_BENCH_ARGS > $out/${name}_$r.json 2> $out/${name}_$r.err
    python - $out/${name}_$r.json $name $r <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "round", sys.argv[3], d["ms_per_step"], "ms/step", d["value"], "img/s")
except Exception as e:
    print(sys.argv[2], "round", sys.argv[3], "FAILED", e)
PY
  done
done
