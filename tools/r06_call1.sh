#!/bin/bash
# round 6, GPU call 1: quick parity subset, de-phase sweep (pp_bench), weight-gradient tile order A/B, teacher dtype A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "distill_loss or wgrad or conv_family" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -3 $O/pytest_subset.log
for d in 0 6 12 18 65548; do
  echo "== THEIA_PP_DEPHASE=$d" >> $O/dephase.txt
  THEIA_PP_DEPHASE=$d timeout 300 build/pp_bench time 30 >> $O/dephase.txt 2>&1
done
THEIA_PP_DEPHASE=12 timeout 300 build/pp_bench check > $O/dephase_check.txt 2>&1; tail -1 $O/dephase_check.txt
AB_BENCH_ARGS="" bash tools/ab_env.sh $O/ab 2 "ctile:THEIA_WGRAD_XCD=1" "tap:THEIA_WGRAD_XCD=tap" "dephase12:THEIA_PP_DEPHASE=12" > $O/ab.txt 2>&1
AB_BENCH_ARGS="--teacher-dtype fp32" bash tools/ab_env.sh $O/ab_fp32 1 "fp32tgt:THEIA_WGRAD_XCD=1" >> $O/ab.txt 2>&1
cat $O/ab.txt
THEIA_BENCH_GEMM_TABLE=1 python bench.py --steps 5 --warmup 2 --no-selfcheck --no-cpu-baseline > $O/table_ctile.json 2> $O/table_ctile.err
THEIA_WGRAD_XCD=tap THEIA_BENCH_GEMM_TABLE=1 python bench.py --steps 5 --warmup 2 --no-selfcheck --no-cpu-baseline > $O/table_tap.json 2> $O/table_tap.err
grep "gemm_wgrad(isolated)" $O/table_ctile.err | head -30
echo ---
grep "gemm_wgrad(isolated)" $O/table_tap.err | head -30
