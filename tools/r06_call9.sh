#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c9; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
THEIA_TEST_DP2_ONE_GPU=1 timeout 600 python -m pytest tests/test_parallel_gpu.py -x -q -s -k "captured_halves_with_the_gradient_exchange_two_ranks or dp2_on_one_gpu" > $O/dp2_one_gpu.log 2>&1; echo "rc=$?" >> $O/dp2_one_gpu.log
grep -E "captured halves|passed|failed|rc=|Error" $O/dp2_one_gpu.log | tail -8
python bench.py --backbone facebook/deit-small-patch16-224 --teachers cdiv --batch 16 --steps 50 --warmup 5 --no-cpu-baseline --no-selfcheck --no-roofline --graph > $O/b16_graph.json 2>/dev/null; cut -c1-220 $O/b16_graph.json
