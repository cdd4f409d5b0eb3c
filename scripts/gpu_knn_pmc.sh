#!/bin/bash
# Matcher evidence: kernel trace, then one SQ counter pass (MFMA busy cycles, MFMA
# instruction count, busy cycles), over 3 launches of 262144 x 1048576.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $O/knn_kt $O/knn_pmc
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/knn_kt -o x -- python3 scripts/prof_knn.py 262144 1048576 > $O/knn_kt.log 2>&1 || { tail -20 $O/knn_kt.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/knn_pmc -o x -- python3 scripts/prof_knn.py 262144 1048576 > $O/knn_pmc.log 2>&1 || { tail -20 $O/knn_pmc.log; exit 1; }
python scripts/pmc_summary.py $O/knn_pmc > $O/knn_pmc.txt
cat $O/knn_kt/x_kernel_stats.csv | cut -c1-200
cat $O/knn_pmc.txt
