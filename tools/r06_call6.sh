#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
AB_BENCH_ARGS="--backbone facebook/deit-small-patch16-224 --batch 256" bash tools/ab_env.sh $O/ab_small 2 "t384:THEIA_WGRAD_TILE=0" "t256:THEIA_WGRAD_TILE=256" > $O/ab_small.txt 2>&1
cat $O/ab_small.txt
THEIA_BENCH_GEMM_TABLE=1 python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 5 --warmup 2 --no-selfcheck --no-cpu-baseline > $O/table_small.json 2> $O/table_small.err
grep "gemm_wgrad(isolated)" $O/table_small.err | head -30
THEIA_WGRAD_TILE=256 THEIA_BENCH_GEMM_TABLE=1 python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 5 --warmup 2 --no-selfcheck --no-cpu-baseline > $O/table_small256.json 2> $O/table_small256.err
grep "gemm_wgrad(isolated)" $O/table_small256.err | head -30
