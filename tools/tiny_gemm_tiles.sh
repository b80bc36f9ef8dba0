cd $GRAFT_REPO_ROOT
S="50432,576,192;50432,192,192;50432,768,192;50432,192,768;50432,192,576"
for t in 0 128128 256256 320256; do echo "== tile $t"; python tools/gemm_bench.py --what nt --mnk "$S" --tile $t --iters 30 2>&1 | grep -v "^$" | tail -6; done
