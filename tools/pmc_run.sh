#!/bin/bash
# rocprofv3 PMC passes over any command (one counter group per pass, kernel-trace only), summarised per kernel.
# usage: tools/pmc_run.sh <outdir> <cmd...>
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- "${CMD[@]}" > $OUT/$name.log 2>&1
}
CMD=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
run tcc FETCH_SIZE
run tcw WRITE_SIZE
python ${GRAFT_REPO_ROOT:-/root/repo}/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
