#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/ab_env.sh $O/ab 2 "auto:THEIA_WGRAD_GROUP=1" "off:THEIA_PP_DEPHASE=0" > $O/ab.txt 2>&1
cat $O/ab.txt
build/pp_bench check > $O/pp_check.txt 2>&1; tail -1 $O/pp_check.txt
