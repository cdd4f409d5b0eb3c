#!/bin/bash
# quick GPU loop: extraction parity tests, then a short bench
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_extract.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
rc=$?; tail -3 gpurun_out/gputests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/gputests.log; exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 --no-cpu "$@" > gpurun_out/bench_quick.log 2>&1 || { tail -20 gpurun_out/bench_quick.log; exit 1; }
python scripts/bench_summary.py gpurun_out/bench_quick.log
