#!/bin/bash
# kernel trace of two 16-frame extraction batches; per-dispatch listing of the second
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $O/kt
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o x -- python3 scripts/prof_extract.py 16 > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
python scripts/kt_summary.py $O/kt > $O/kt.txt
grep -E "${1:-.}" $O/kt.txt
