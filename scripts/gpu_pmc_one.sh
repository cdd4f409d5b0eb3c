#!/bin/bash
# One rocprofv3 --pmc pass over two 16-frame 1080p extraction batches.  $1 = tag, $2 = counters (quoted).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $O/pmc_$1
timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -o x -- python3 scripts/prof_extract.py 16 > $O/pmc_$1.log 2>&1 || { tail -20 $O/pmc_$1.log; exit 1; }
python scripts/pmc_summary.py $O/pmc_$1 > $O/pmc_$1.txt
grep -A9 -E "^k_(descriptors|desc_order)" $O/pmc_$1.txt
