"""CPU tests of the host side: the C ABI library loads and exports everything
include/sift_amd.h declares, host-only helpers are right, and GPU entry points
fail loudly (never fall back to a CPU path) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sift_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sift_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sift_lib):
    lib = ctypes.CDLL(sift_lib.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libsift_amd.so does not export %s" % name
    assert sorted(sift_lib.EXPORTED_SYMBOLS) == declared


def test_keypoint_struct_layout_matches_cv_keypoint(sift_lib):
    # cv::KeyPoint: Point2f pt; float size, angle, response; int octave, class_id
    dt = sift_lib.KEYPOINT_DTYPE
    assert dt.itemsize == 28
    assert [dt.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]
    assert sift_lib.KNN2_DTYPE.itemsize == 16


def test_octave_geometry(sift_lib, known):
    # sift.cpp:36: round(log2(min(w,h))) on the pre-upsample size
    assert sift_lib.num_octaves(512, 512) == 9
    assert sift_lib.num_octaves(1920, 1080) == 10
    assert sift_lib.num_octaves(3840, 2160) == 11
    assert sift_lib.num_octaves(23, 900) == 5      # log2(23) = 4.52 -> 5
    assert sift_lib.num_octaves(22, 900) == 4      # log2(22) = 4.46 -> 4
    dims = [list(sift_lib.octave_dims(1920, 1080, o)) for o in range(10)]
    assert dims == known["octave_sizes_1920x1080"]   # odd sizes halve with round-half-to-even
    assert sum(w * h for w, h in dims) == 11059262
    dims4k = [sift_lib.octave_dims(3840, 2160, o) for o in range(11)]
    assert sum(w * h for w, h in dims4k) == 44236862


def test_dead_leaves_generator_known_answer(sift_lib, known):
    img = sift_lib.synth_dead_leaves(1, 0, 1920, 1080)
    assert int(img.sum(dtype=np.int64)) == known["dead_leaves_1920x1080_seed1_frame0"]["pixel_sum"]
    assert np.array_equal(img, sift_lib.synth_dead_leaves(1, 0, 1920, 1080))
    assert not np.array_equal(img, sift_lib.synth_dead_leaves(1, 1, 1920, 1080))


def test_dead_leaves_generator_against_python_restatement(sift_lib):
    """Independent pure-Python version of the generator on a small frame."""
    M = (1 << 64) - 1
    w, h, seed, frame = 96, 64, 3, 5
    state = (seed * 0x9E3779B97F4A7C15 + frame) & M

    def nxt():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & M
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    img = np.full((h, w), 128, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(w * h // 128):
        cx, cy, e, g = nxt() % w, nxt() % h, nxt() % 64, nxt() % 256
        r = 2 + e ** 3 // 4096
        img[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = g
    assert np.array_equal(img, sift_lib.synth_dead_leaves(seed, frame, w, h))


def test_synthetic_descriptors(sift_lib):
    tr = sift_lib.synth_descriptors(5, 5000)
    assert tr.shape == (5000, 128) and tr.max() <= 102
    assert np.array_equal(tr, sift_lib.synth_descriptors(5, 5000))
    assert np.array_equal(tr[:100], sift_lib.synth_descriptors(5, 100))   # rows are independent of n
    q = sift_lib.synth_queries(6, tr, 1000)
    assert q.shape == (1000, 128) and q.max() <= 102
    # even queries are noisy copies of some train row: nearest train row within 128*4
    d = ((q[0].astype(np.int32) - tr.astype(np.int32)) ** 2).sum(1)
    assert d.min() <= 128 * 4
    d = ((q[1].astype(np.int32) - tr.astype(np.int32)) ** 2).sum(1)
    assert d.min() > 128 * 4


def test_min_dist_filter_matches_oracle(sift_lib, oracle):
    rng = np.random.default_rng(0)
    knn = np.zeros(200, sift_lib.KNN2_DTYPE)
    knn["d1"] = rng.integers(0, 40000, 200)
    knn["i1"] = rng.integers(0, 1000, 200)
    good = sift_lib.min_dist_filter(knn)
    ref = oracle.min_dist_filter(np.sqrt(knn["d1"].astype(np.float32)))
    assert good.tolist() == ref.tolist()
    # all distances above 100: min_dist keeps its initial value 100 (main.cpp:77)
    knn["d1"] = rng.integers(150 ** 2, 400 ** 2, 200)
    good = sift_lib.min_dist_filter(knn)
    ref = oracle.min_dist_filter(np.sqrt(knn["d1"].astype(np.float32)))
    assert good.tolist() == ref.tolist()
    assert all(np.sqrt(np.float32(knn["d1"][i])) < 300 for i in good)


def test_bad_arguments_return_error_codes(sift_lib):
    L = sift_lib.lib()
    assert L.sift_ctx_create(0, None) == sift_lib.api.SIFT_ERR_INVALID
    assert b"NULL" in L.sift_last_error()
    assert L.sift_batch_create(None, 64, 64, 1, 0, None) == sift_lib.api.SIFT_ERR_INVALID
    assert L.sift_batch_run(None, 1) == sift_lib.api.SIFT_ERR_INVALID
    assert L.sift_knn2_u8(None, None, 1, None, 1, None, None) == sift_lib.api.SIFT_ERR_INVALID
    ow, oh = ctypes.c_int(), ctypes.c_int()
    assert L.sift_octave_dims(0, 10, 0, ctypes.byref(ow), ctypes.byref(oh)) == sift_lib.api.SIFT_ERR_INVALID
    assert L.sift_ctx_destroy(None) == 0 and L.sift_batch_destroy(None) == 0 and L.sift_result_free(None) == 0


def test_no_cpu_fallback_without_gpu(sift_lib):
    """Without a GPU the engine must refuse to run, not silently compute elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sift_lib.SiftError) as e:
        sift_lib.Context(0)
    assert e.value.code in (sift_lib.api.SIFT_ERR_HIP, sift_lib.api.SIFT_ERR_INVALID)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under sift_amd/ or include/ may reference it."""
    for base in ("sift_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_fast_atan2_is_within_two_ulps_of_libm(tmp_path):
    """sift_amd/csrc/atan2_fast.h compiled for the host (same fma sequence as on the GPU, the
    hardware reciprocal estimate replaced by a float one) against glibc's atan2, and its
    division by pi against the IEEE division of (a*180)/PI (sift.cpp:415)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "atan2_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "sift_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "atan2_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "div_pi mismatch" not in out
    assert "bin_mismatches 0" in out  # div_360 and round_half_away reproduce (ori*36)/360 and round() (sift.cpp:416)
    m = re.search(r"samples (\d+) max_ulp ([\d.]+) not_identical (\d+) max_ulp_deg ([\d.]+)", out)
    assert m, out
    assert int(m.group(1)) > 4_000_000
    assert float(m.group(2)) <= 2.0
    assert float(m.group(4)) <= 3.0
