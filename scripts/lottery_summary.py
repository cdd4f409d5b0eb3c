"""Groups the dispatches of one kernel from a rocprofv3 --kernel-trace --pmc run of scripts/dev_alloc_lottery.py by
workspace allocation (every allocation runs the pipeline three times; launches are numbered in dispatch order) and
prints mean duration and counters per allocation."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_blur_march<5>"
runs_per_alloc = 3
dur = {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
cnt = collections.defaultdict(dict)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(set(dur) | set(cnt))
# the first launch of every pipeline run is the octave-0 launch: the largest duration among each run's launches
per_run = len(ids) // (6 * runs_per_alloc) if ids else 0
print("dispatches of %s: %d (%d per pipeline run)" % (kernel, len(ids), per_run))
if per_run:
    for a in range(len(ids) // (per_run * runs_per_alloc)):
        chunk = ids[a * per_run * runs_per_alloc:(a + 1) * per_run * runs_per_alloc]
        big = [i for k, i in enumerate(chunk) if k % per_run == 0]   # octave 0
        names = sorted({n for i in big for n in cnt.get(i, {})})
        line = "alloc %d  octave-0 launch %.1f us" % (a, sum(dur.get(i, 0) for i in big) / len(big))
        for n in names:
            line += "  %s %.3g" % (n.replace("TCC_", "").replace("_sum", ""), sum(cnt[i].get(n, 0) for i in big) / len(big))
        print(line)
