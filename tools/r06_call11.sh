#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c11; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad or conv_family" > $O/pytest.log 2>&1; tail -1 $O/pytest.log
bash tools/ab_env.sh $O/ab 2 "r:THEIA_WGRAD_ISSUE=r" "split:THEIA_WGRAD_ISSUE=split" > $O/ab.txt 2>&1; cat $O/ab.txt
AB_BENCH_ARGS="--backbone facebook/deit-small-patch16-224 --batch 256" bash tools/ab_env.sh $O/ab_small 2 "r:THEIA_WGRAD_ISSUE=r" "split:THEIA_WGRAD_ISSUE=split" > $O/ab_small.txt 2>&1; cat $O/ab_small.txt
