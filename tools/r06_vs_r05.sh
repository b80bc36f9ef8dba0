#!/bin/bash
# round 5 (commit e859bd4, worktree build/r05_tree with its own library) against the current tree, arms alternating on one box
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/vs_r05; mkdir -p $O
run() {  # name dir round args...
  n=$1; d=$2; r=$3; shift 3
  (cd $d && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-selfcheck --no-roofline "$@" > $O/${n}_$r.json 2> $O/${n}_$r.err)
  python - $O/${n}_$r.json $n $r <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "round", sys.argv[3], d["ms_per_step"], "ms/step", d["value"], "img/s")
except Exception as e:
    print(sys.argv[2], "round", sys.argv[3], "FAILED", e)
PY
}
for r in 1 2 3; do
  run base_r05 build/r05_tree $r
  run base_r06 . $r
  run base_r06_fp32_targets . $r --teacher-dtype fp32
done
for r in 1 2; do
  run small_r05 build/r05_tree $r --backbone facebook/deit-small-patch16-224 --batch 256
  run small_r06 . $r --backbone facebook/deit-small-patch16-224 --batch 256
  run tiny_r05 build/r05_tree $r --backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256
  run tiny_r06 . $r --backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256
done
