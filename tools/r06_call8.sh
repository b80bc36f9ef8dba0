#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c8; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
THEIA_BENCH_GEMM_TABLE=1 python bench.py --steps 10 --warmup 3 --no-selfcheck --no-cpu-baseline > $O/table.json 2> $O/table.err
grep "gemm_wgrad(isolated)" $O/table.err | head -30; cut -c1-150 $O/table.json
THEIA_BENCH_GEMM_TABLE=1 python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 10 --warmup 3 --no-selfcheck --no-cpu-baseline > $O/table_small.json 2> $O/table_small.err
grep "gemm_wgrad(isolated)" $O/table_small.err | head -30;  cut -c1-150 $O/table_small.json
grep "gemm_nt(isolated)" $O/table_small.err | head -40
