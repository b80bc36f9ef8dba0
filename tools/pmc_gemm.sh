#!/bin/bash
# rocprofv3 PMC passes over the GEMM micro-benchmark (one counter group per pass, kernel-trace only).
# usage: tools/pmc_gemm.sh <outdir> [gemm_bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python $REPO/tools/gemm_bench.py --iters 3 "${ARGS[@]}" > $OUT/$name.log 2>&1
}
ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
find $OUT -name "*.csv" | head -20
