#!/bin/bash
# kernel statistics of the homography check script
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $O/prof_h
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h -o h -- python3 scripts/dev_homography_check.py > $O/prof_h.log 2>&1 || { tail -20 $O/prof_h.log; exit 1; }
cat $O/prof_h/h_kernel_stats.csv
