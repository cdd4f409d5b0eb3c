#!/bin/bash
# A/B of library variants inside one GPU call: every variants/lib_*.so is copied over the library in turn and
# the per-stage device times of three fresh workspaces are printed (placement differs between workspaces).
set -o pipefail
cp sift_amd/libsift_amd.so /tmp/lib_keep.so
for v in variants/lib_*.so; do
  cp $v sift_amd/libsift_amd.so
  echo "== $v"
  timeout -k 10 200 python -u scripts/dev_alloc_lottery.py ${1:-64} 3 2>&1 | grep -v amdgpu.ids || exit 1
done
cp /tmp/lib_keep.so sift_amd/libsift_amd.so
