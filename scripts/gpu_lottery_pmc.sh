#!/bin/bash
# Why do the memory-bound blur levels run 2-3 % faster in some allocations of the workspace than in others?
# Six allocations of a 16-frame 1080p workspace in one process under rocprofv3 with per-dispatch durations and the
# L2's memory-side stall counters; scripts/lottery_summary.py groups the level-1 blur dispatches by allocation.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $O/lottery
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/lottery -o x -- python3 scripts/dev_alloc_lottery.py 16 6 > $O/lottery.log 2>&1 || { tail -20 $O/lottery.log; exit 1; }
grep -E "^[0-9] " $O/lottery.log | cut -c1-120
python scripts/lottery_summary.py $O/lottery | tee $O/lottery_summary.txt
