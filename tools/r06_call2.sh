#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
for d in 0 24 32 131084 131090 131096 131104 131120 0; do
  echo "== THEIA_PP_DEPHASE=$d" >> $O/dephase.txt
  THEIA_PP_DEPHASE=$d timeout 300 build/pp_bench time 30 2>&1 | grep "fc1 \|qkv \|fc2_d\|fc2  \|up64c" | cut -c1-108 >> $O/dephase.txt
done
cat $O/dephase.txt
