#!/bin/bash
# extraction parity tests + quick bench + one FETCH_SIZE pass over two 16-frame batches
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
bash scripts/gpu_check.sh "$@" || exit 1
rm -rf $O/pmc_fetch
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o x -- python3 scripts/prof_extract.py 16 > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
python scripts/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch.txt
grep -A1 -E "^k_" $O/pmc_fetch.txt | grep -v "^--" | paste - - | awk '{printf "%-22s %14.0f KiB x2 = %8.2f GB per batch\n", $1, $3, $3*2*1024/2/1e9}'
