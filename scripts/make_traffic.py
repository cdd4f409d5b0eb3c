"""Turns the two rocprofv3 --pmc passes of scripts/gpu_full.sh (FETCH_SIZE, WRITE_SIZE over
scripts/prof_extract.py) into profiles/traffic.json: HBM bytes per launch of every kernel.

gfx950 corrections (calibrated by scripts/ubench/copy8.hip in the same script: a streaming
copy of known size with this engine's 8-byte-per-lane accesses): FETCH_SIZE reports half of
the bytes read, WRITE_SIZE the bytes written; both are in KiB.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import source_hash  # noqa: E402  (identity of the kernel sources the counters were measured on)

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64


def collect(root, counter):
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("sa::(anonymous namespace)::", "").replace("void ", ""))
            tot[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


fetch, nf = collect(out_dir + "/pmc_fetch", "FETCH_SIZE")
write, nw = collect(out_dir + "/pmc_write", "WRITE_SIZE")
cal_f, ncf = collect(out_dir + "/pmc_cal_fetch", "FETCH_SIZE")
cal_w, ncw = collect(out_dir + "/pmc_cal_write", "WRITE_SIZE")
known_kib = 1024 * 1024.0  # copy8 1024: 1 GiB read and 1 GiB written per launch
fetch_factor = known_kib / (cal_f["k_copy8"] / ncf["k_copy8"])
write_factor = known_kib / (cal_w["k_copy8"] / ncw["k_copy8"])
res = {"_calibration": {"fetch_factor": fetch_factor, "write_factor": write_factor,
                        "note": "known bytes / reported bytes of an 8-byte-per-lane streaming copy of 1 GiB"},
       "_frames_per_launch": frames, "_source_hash": source_hash(),
       "_unit": "HBM bytes per launch (mean over the launches of two %d-frame 1080p batches)" % frames}
names = {"k_blur_march<5>": "blur_l1", "k_blur_march<6>": "blur_l2", "k_blur_march<8>": "blur_l3",
         "k_blur_march<10>": "blur_l4", "k_blur_pair<5, 6>": "blur_l12", "k_blur_march<13>": "blur_l5", "k_base_march<5>": "base",
         "k_descriptors<false, 2>": "k_descriptors", "k_orient<false>": "k_orient", "k_rank_sort<false>": "k_rank_sort"}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    rd = fetch.get(k, 0.0) * 1024 * fetch_factor / max(nf.get(k, 1), 1)
    wr = write.get(k, 0.0) * 1024 * write_factor / max(nw.get(k, 1), 1)
    key = names.get(k, k)
    res[key] = rd + wr
    res[key + "_detail"] = {"kernel": k, "read": rd, "written": wr, "launches_sampled": nf.get(k, 0)}
json.dump(res, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.endswith("_detail")}, indent=1))
