#!/bin/bash
# Counter evidence for the per-keypoint kernels (k_descriptors, k_orient) and the fused scan: four separate
# rocprofv3 --pmc passes (instruction issue and waits, instruction mix and LDS conflicts, L2 hit/miss and fabric
# reads, L1) over two 16-frame 1080p extraction batches.  $1 = tag of the output directory.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-desc}
export TMPDIR=/tmp
cd $R
pass() {
  rm -rf $O/pmc_${T}_$1
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_${T}_$1 -o x -- python3 scripts/prof_extract.py 16 > $O/pmc_${T}_$1.log 2>&1 || { tail -20 $O/pmc_${T}_$1.log; exit 1; }
  echo "== pass $1: $2"; python scripts/pmc_summary.py $O/pmc_${T}_$1 | grep -A12 -E "^k_(descriptors|orient|extrema_grad|desc_order)"
}
{
pass a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" &&
pass b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" &&
pass c "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" &&
pass d "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
} > $O/pmc_${T}.txt 2>&1
rc=$?
tail -60 $O/pmc_${T}.txt
exit $rc
