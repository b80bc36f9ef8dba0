#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
bash tools/ab_env.sh $O/ab 2 "group:THEIA_WGRAD_GROUP=1" "nogroup:THEIA_WGRAD_GROUP=0" "group_deph:THEIA_WGRAD_GROUP=1 THEIA_PP_DEPHASE=131090" > $O/ab.txt 2>&1
cat $O/ab.txt
THEIA_BENCH_GEMM_TABLE=1 python bench.py --steps 5 --warmup 2 --no-selfcheck --no-cpu-baseline > $O/table.json 2> $O/table.err
grep "gemm_wgrad(isolated)" $O/table.err | head -30
