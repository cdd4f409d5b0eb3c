cp sift_amd/libsift_amd.so /tmp/lib_keep.so
for v in variants/lib_*.so; do cp $v sift_amd/libsift_amd.so; echo "== $v"; timeout -k 10 100 python -u scripts/dev_desc_work.py 2>&1 | grep -v amdgpu.ids || exit 1; done
cp /tmp/lib_keep.so sift_amd/libsift_amd.so
