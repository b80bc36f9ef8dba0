cd /root/repo; mkdir -p gpurun_out/r2c; rm -f gpurun_out/r2c/*
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "cast_batch" 2>&1 | tail -5 > gpurun_out/r2c/tests.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5 >> gpurun_out/r2c/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof -o p -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-selfcheck --no-roofline > /root/repo/gpurun_out/r2c/bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python /root/repo/tools/rocpd_stats.py $DB --csv /root/repo/gpurun_out/r2c/stats.csv > /dev/null 2>&1
