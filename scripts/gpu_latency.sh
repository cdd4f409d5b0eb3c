set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_extract.py tests/test_cpp_dropin.py tests/test_stitch.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1; rc=$?; tail -5 gpurun_out/t.log; [ $rc -eq 0 ] || exit $rc
for s in "512 512" "1920 1080"; do timeout -k 5 60 scripts/ubench/extract_latency $s || exit 1; timeout -k 5 60 scripts/ubench/handler_latency $s || exit 1; done
