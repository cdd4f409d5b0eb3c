#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace, 64-frame batches) for each library variant
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
cp sift_amd/libsift_amd.so /tmp/lib_keep.so
for v in variants/lib_*.so; do
  cp $v sift_amd/libsift_amd.so
  t=$(basename $v .so)
  rm -rf $O/kt_ab_$t
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ab_$t -o x -- python3 scripts/prof_extract.py 64 > $O/kt_ab_$t.log 2>&1 || { tail -20 $O/kt_ab_$t.log; exit 1; }
  echo "== $v"; cut -d, -f1-4 $O/kt_ab_$t/x_kernel_stats.csv | grep -E "${1:-k_desc|k_orient|k_rank}"
done
cp /tmp/lib_keep.so sift_amd/libsift_amd.so
