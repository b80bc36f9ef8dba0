#!/bin/bash
# whole-step A/B of prebuilt libraries on ONE box, arms alternating: tools/ab_libs.sh <outdir> <rounds> lib1.so lib2.so ...
out=$1; rounds=$2; shift 2
mkdir -p $out
keep=theia_amd/lib/libtheia_hip.so.keep
cp theia_amd/lib/libtheia_hip.so $keep
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    cp $lib theia_amd/lib/libtheia_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-selfcheck --no-roofline $AB_BENCH_ARGS > $out/${name}_$r.json 2> $out/${name}_$r.err
    python - $out/${name}_$r.json $name $r <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "round", sys.argv[3], d["ms_per_step"], "ms/step", d["value"], "img/s")
PY
  done
done
cp $keep theia_amd/lib/libtheia_hip.so; rm -f $keep
