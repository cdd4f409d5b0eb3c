"""The bench line: the last recorded run (profiles/r3_bench.json, written by `python bench.py` on an MI355X)
carries every field the measurement contract names, with consistent values."""
import json
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def load(name):
    lines = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()
    return json.loads(lines[-1])


def test_recorded_bench_line_has_the_contract_fields():
    d = load("r3_bench.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    frames = d["config"]["frames_per_gpu_per_step"] * d["steps"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # the counters see the bytes the kernel moves (no stored difference-of-Gaussians level): 2/3 of the algorithmic figure
    m = r["moved"]
    assert abs(m["bytes_per_launch"] * 1.5 - r["algorithmic_bytes_per_launch"]) < 1e-6 * m["bytes_per_launch"]
    assert r["traffic"] is None or 0.95 < r["traffic"] / m["bytes_per_launch"] < 1.25
    for k in ("extrema_gradients", "descriptor", "base"):
        e = d["kernels"][k]
        assert e["bound"] and 0 < e["frac"] < 1 and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-9
    kn = d["knn"]
    assert (kn["queries"], kn["train"]) == (1 << 20, 1310720)          # one GPU's share of 1 M x 10 M
    assert kn["roofline"]["bound"] == "mfma" and kn["roofline"]["peak"] == 5000
    kc = kn["cpu_baseline"]
    assert kc["unit"] == "pairs/s" and kc["cores"] >= 1 and kc["value"] > 0 and kc["gpu_results_identical_on_sample"] is True
    assert d["end_to_end"]["frames_per_s"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    # the other benchmark configurations ride in the same line
    k4 = d["configs"]["3840x2160_batch"]
    assert k4["octaves"] == 11 and k4["frames_per_s"] > 0 and 0 < k4["roofline"]["frac"] < 1
    assert 0 < k4["pyramid"]["moved_frac_of_hbm_peak"] < k4["pyramid"]["frac_of_hbm_peak"] < 1
    one = d["configs"]["single_2048x1080_frame"]
    assert one["host_to_host_ms_median"] > 0 and len(one["stage_ms"]) == 5
    oc = one["cpu_baseline"]
    assert oc["cores"] >= 1 and oc["value"] > one["host_to_host_ms_median"] and oc["gpu_keypoint_count_equal"] is True


def test_rocprof_summary_agrees_with_the_event_timing():
    """profiles/r3_bench_main_kernel_stats.csv is the rocprofv3 --kernel-trace --stats summary of the run that wrote
    r3_bench_profiled_main.json (`bench.py --no-configs`: the default run without the legs of the other
    configurations, whose launches of other shapes would enter the profiler's per-kernel average): the dominant
    kernel's average duration must match the HIP-event figure."""
    import csv
    d = load("r3_bench_profiled_main.json")
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r3_bench_main_kernel_stats.csv"))))
    name = d["roofline"]["kernel"].split()[0]   # k_blur_march<13>
    hit = [r for r in rows if name in r["Name"]]
    assert len(hit) == 1
    avg_ms = float(hit[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - d["roofline"]["avg_launch_ms"]) / avg_ms < 0.05
