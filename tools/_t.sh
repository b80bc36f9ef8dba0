cd /root/repo; mkdir -p gpurun_out/r2e; rm -f gpurun_out/r2e/*
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "linear or conv or pad or patch or bench_size or fp8" 2>&1 | tail -4 > gpurun_out/r2e/tests.log
for v in old new old new; do
  cp build/libtheia_hip_$v.so theia_amd/lib/libtheia_hip.so
  echo "# $v" >> gpurun_out/r2e/ab_bench.txt
  THEIA_BENCH_GEMM_TABLE=1 timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-selfcheck 2> gpurun_out/r2e/table_$v.txt | tail -1 >> gpurun_out/r2e/ab_bench.txt
done
cp build/libtheia_hip_new.so theia_amd/lib/libtheia_hip.so
