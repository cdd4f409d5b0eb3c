#!/bin/bash
# Where the matcher's waves wait: two SQ counter passes over 3 launches of 262144 x 1048576
# (issue/wait split; instruction mix and LDS activity).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
pass() {
  rm -rf $O/knn_pmc_$1
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/knn_pmc_$1 -o x -- python3 scripts/prof_knn.py 262144 1048576 > $O/knn_pmc_$1.log 2>&1 || { tail -20 $O/knn_pmc_$1.log; exit 1; }
  echo "== pass $1: $2"; python scripts/pmc_summary.py $O/knn_pmc_$1 | grep -A10 "^k_knn2"
}
{
pass a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" &&
pass b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" &&
pass c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES"
} > $O/knn_pmc2.txt 2>&1
rc=$?
cat $O/knn_pmc2.txt
exit $rc
