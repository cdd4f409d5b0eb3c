#!/bin/bash
# fabric reads of the per-keypoint kernels for each library variant
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
cp sift_amd/libsift_amd.so /tmp/lib_keep.so
for v in variants/lib_*.so; do
  cp $v sift_amd/libsift_amd.so
  t=$(basename $v .so)
  rm -rf $O/pmc_ab_$t
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_ab_$t -o x -- python3 scripts/prof_extract.py 16 > $O/pmc_ab_$t.log 2>&1 || { tail -20 $O/pmc_ab_$t.log; exit 1; }
  echo "== $v"; python scripts/pmc_summary.py $O/pmc_ab_$t | grep -A5 -E "^k_(descriptors|desc_order)"
done
cp /tmp/lib_keep.so sift_amd/libsift_amd.so
