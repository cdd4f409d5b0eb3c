#!/bin/bash
set -o pipefail
cp sift_amd/libsift_amd.so /tmp/lib_keep.so
for v in variants/lib_*.so; do
  cp $v sift_amd/libsift_amd.so
  echo "== $v"
  timeout -k 10 400 python -u -m pytest tests/test_gpu_extract.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
  timeout -k 10 200 python -u scripts/dev_alloc_lottery.py 64 3 2>&1 | grep -v amdgpu.ids || exit 1
done
cp /tmp/lib_keep.so sift_amd/libsift_amd.so
