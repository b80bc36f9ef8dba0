cd /root/repo; mkdir -p gpurun_out/r2a; rm -f gpurun_out/r2a/*
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -3 > gpurun_out/r2a/tests.log
for v in old new old new; do
  cp build/libtheia_hip_$v.so theia_amd/lib/libtheia_hip.so
  echo "# $v" >> gpurun_out/r2a/attn.txt
  timeout 120 python tools/attn_bench.py --iters 50 2>&1 | grep -v amdgpu >> gpurun_out/r2a/attn.txt
done
cp build/libtheia_hip_new.so theia_amd/lib/libtheia_hip.so
