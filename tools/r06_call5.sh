#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
AB_BENCH_ARGS="--backbone facebook/deit-small-patch16-224 --batch 256" bash tools/ab_env.sh $O/ab_small 2 "auto:THEIA_WGRAD_GROUP=auto" "pair:THEIA_WGRAD_GROUP=1" "none:THEIA_WGRAD_GROUP=0" > $O/ab_small.txt 2>&1
cat $O/ab_small.txt
AB_BENCH_ARGS="--backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256" bash tools/ab_env.sh $O/ab_tiny 2 "auto:THEIA_WGRAD_GROUP=auto" "pair:THEIA_WGRAD_GROUP=1" "none:THEIA_WGRAD_GROUP=0" > $O/ab_tiny.txt 2>&1
cat $O/ab_tiny.txt
bash tools/ab_env.sh $O/ab_base 1 "auto:THEIA_WGRAD_GROUP=auto" "all:THEIA_WGRAD_GROUP=all" > $O/ab_base.txt 2>&1
cat $O/ab_base.txt
