#!/bin/bash
# profile-only tail of gpu_full.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
sed -n '/^rm -rf/,$p' scripts/gpu_full.sh > /tmp/prof_tail.sh
R=$R O=$O bash /tmp/prof_tail.sh
