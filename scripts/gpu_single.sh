#!/bin/bash
# Single-image path: parity tests, kernel timeline of one 512x512 extraction (rocprofv3 kernel trace), wall times.
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_extract.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
rm -rf gpurun_out/kt_single
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_single -o x -- python3 scripts/prof_single.py ${1:-512} ${2:-512} > gpurun_out/kt_single.log 2>&1 || { tail gpurun_out/kt_single.log; exit 1; }
python scripts/kt_timeline.py gpurun_out/kt_single 4 | cut -c1-60 > gpurun_out/kt_single.txt
grep -v "q3 " gpurun_out/kt_single.txt | tail -${3:-8}
timeout -k 10 100 python scripts/dev_single_latency.py 2>&1 | grep -v amdgpu.ids
