#!/bin/bash
# Evidence run on the GPU box: full GPU test suite, smoke, the default
# bench under rocprofv3 --kernel-trace --stats --output-format csv, then HBM byte counters
# (separate passes) for the extraction kernels and the 8-byte copy calibration.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gputests_full.log 2>&1
rc=$?; tail -3 $O/gputests_full.log
if [ $rc -ne 0 ]; then tail -60 $O/gputests_full.log; exit $rc; fi
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write $O/pmc_cal_fetch $O/pmc_cal_write
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o x -- python3 scripts/prof_extract.py 64 > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o x -- python3 scripts/prof_extract.py 64 > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_cal_fetch -o x -- scripts/ubench/copy8 1024 > $O/pmc_cal_fetch.log 2>&1 || { tail -20 $O/pmc_cal_fetch.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_cal_write -o x -- scripts/ubench/copy8 1024 > $O/pmc_cal_write.log 2>&1 || { tail -20 $O/pmc_cal_write.log; exit 1; }
bash scripts/gpu_desc_pmc.sh final > $O/desc_pmc_run.log 2>&1 || { tail -20 $O/desc_pmc_run.log; exit 1; }
for d in pmc_fetch pmc_write pmc_cal_fetch pmc_cal_write; do echo "== $d"; python scripts/pmc_summary.py $O/$d; done > $O/pmc_all.txt
python scripts/make_traffic.py $O 64 > $O/traffic_summary.txt && cp profiles/traffic.json $O/traffic.json
tail -12 $O/traffic_summary.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py > $O/bench_profiled.json 2> $O/bench_profiled.err || { tail -20 $O/bench_profiled.err; exit 1; }
python scripts/bench_summary.py $O/bench_profiled.json
# the same under the profiler without the legs of the other configurations and without the host-to-host leg (whose
# first and last launches are a quarter of the size): every launch of a kernel then has the shape the roofline figure
# is quoted for, and the profiler's per-kernel average can be held against the event timing
rm -rf $O/prof_bench_main
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_main -o bench -- python3 bench.py --no-configs --no-e2e > $O/bench_profiled_main.json 2> $O/bench_profiled_main.err || { tail -20 $O/bench_profiled_main.err; exit 1; }
python scripts/bench_summary.py $O/bench_profiled_main.json
timeout -k 10 300 python -u bench.py > $O/bench_plain.json 2> $O/bench_plain.err || { tail -20 $O/bench_plain.err; exit 1; }
python scripts/bench_summary.py $O/bench_plain.json
