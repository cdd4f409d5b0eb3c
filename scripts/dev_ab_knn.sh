#!/bin/bash
# matcher A/B inside one GPU call: every variants/lib_*.so in turn, parity of the small cases first, then timing
set -o pipefail
cp sift_amd/libsift_amd.so /tmp/lib_keep.so
for v in variants/lib_*.so; do
  cp $v sift_amd/libsift_amd.so
  echo "== $v"
  [ -n "$SKIP_PARITY" ] || { timeout -k 10 300 python -u -m pytest tests/test_gpu_knn.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1; }
  timeout -k 10 200 python -u scripts/time_knn.py ${1:-262144} ${2:-1048576} 2>&1 | grep -v amdgpu.ids || exit 1
done
cp /tmp/lib_keep.so sift_amd/libsift_amd.so
