#!/bin/bash
# one SQ counter pass over two 16-frame extraction batches
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $O/pmc_sq
timeout -s KILL 120 rocprofv3 --pmc "$1" --output-format csv -d $O/pmc_sq -o x -- python3 scripts/prof_extract.py 16 > $O/pmc_sq.log 2>&1 || { tail -20 $O/pmc_sq.log; exit 1; }
python scripts/pmc_summary.py $O/pmc_sq > $O/pmc_sq.txt
cat $O/pmc_sq.txt
