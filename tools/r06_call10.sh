#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c10; mkdir -p $O
for lib in split bothr; do
  cp build/alt/lib_$lib.so theia_amd/lib/libtheia_hip.so
  python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad or conv_family" > $O/pytest_$lib.log 2>&1; tail -1 $O/pytest_$lib.log
  THEIA_BENCH_GEMM_TABLE=1 python bench.py --steps 5 --warmup 2 --no-selfcheck --no-cpu-baseline > $O/table_$lib.json 2> $O/table_$lib.err
  echo "== $lib"; grep "gemm_wgrad(isolated)" $O/table_$lib.err | head -8 | cut -c17-
done
cp build/alt/lib_split.so theia_amd/lib/libtheia_hip.so
bash tools/ab_libs.sh $O/ab 2 build/alt/lib_split.so build/alt/lib_bothr.so
