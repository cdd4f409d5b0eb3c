"""findHomography(obj, scene, RANSAC, 3, mask) (main.cpp:105).

The algorithm lives in OpenCV's calib3d, which the reference neither vendors nor
pins and which is not installed here, and the reference holds no homography or
inlier mask: PARITY WITH OPENCV IS UNPINNED.  What is tested: the CPU
restatement (oracle/homography_oracle.c) against independent restatements of
its parts and against ground truth; and, on the GPU, that the HIP path
reproduces the restatement bit for bit (matrix, mask, iteration counts).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

H_TRUE = np.array([[1.1, 0.05, 30.0], [-0.03, 0.95, -12.0], [1e-5, -2e-5, 1.0]])


def project(H, pts):
    p = np.c_[pts, np.ones(len(pts))] @ H.T
    return p[:, :2] / p[:, 2:]


def make_case(seed, n, outliers, noise=0.3, H=H_TRUE, extent=(1920, 1080)):
    rng = np.random.default_rng(seed)
    src = (rng.random((n, 2)) * extent).astype(np.float32)
    dst = (project(H, src) + rng.normal(0, noise, (n, 2))).astype(np.float32)
    k = int(n * outliers)
    dst[:k] = (rng.random((k, 2)) * extent).astype(np.float32)
    return src, dst


CASES = [  # (seed, n, outlier fraction, noise)
    (1, 500, 0.0, 0.3), (2, 500, 0.5, 0.3), (3, 2000, 0.7, 0.3), (4, 5, 0.0, 0.3), (5, 4, 0.0, 0.0),
    (6, 50, 0.9, 0.3), (7, 12, 0.25, 0.0), (8, 10000, 0.6, 0.5), (9, 64, 0.0, 0.0), (10, 65, 0.3, 1.0),
    (11, 257, 0.8, 0.2), (12, 1500, 0.85, 0.3),
]


# ---- the restatement itself (CPU) ---------------------------------------------------

def test_rng_is_the_multiply_with_carry_generator(oracle):
    """cv::RNG::next: state = lo32(state) * 4164903690 + hi32(state); a zero seed becomes 0xffffffff."""
    for seed in (0xFFFFFFFFFFFFFFFF, 12345, 0):
        s = seed if seed else 0xFFFFFFFF
        want = []
        for _ in range(50):
            s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
            want.append(s & 0xFFFFFFFF)
        assert oracle.homography_rng_sequence(seed, 50).tolist() == want


def test_subsets_are_distinct_and_follow_the_generator(oracle):
    src, dst = make_case(21, 40, 0.0)
    subs = oracle.homography_subsets(src, dst, 200)
    assert subs.shape == (200, 4)
    assert all(len(set(row)) == 4 for row in subs.tolist())
    assert subs.min() >= 0 and subs.max() < 40
    # replay: indices are next() % count, redrawn on repeats; whole subsets are redrawn when rejected,
    # so every accepted subset must appear, in order, in the stream of 4-tuples the generator yields
    stream = (oracle.homography_rng_sequence(0xFFFFFFFFFFFFFFFF, 4000) % 40).tolist()
    pos = 0
    for row in subs.tolist():
        while True:
            cand = []
            while len(cand) < 4:
                v = stream[pos]
                pos += 1
                if v not in cand:
                    cand.append(v)
            if cand == row:
                break
    assert pos <= 4000


def test_exact_points_give_the_exact_homography(oracle):
    src, _ = make_case(22, 300, 0.0)
    dst = project(H_TRUE, src.astype(np.float64)).astype(np.float32)
    H, mask, info = oracle.find_homography(src, dst)
    assert info["found"] and mask.all() and info["inliers"] == 300
    err = np.abs(project(H, src) - dst).max()
    assert err < 2e-3  # float32 coordinates of ~2000 px carry 1e-4 px of rounding
    assert np.allclose(H / H[2, 2], H_TRUE, rtol=0, atol=2e-3)


def test_four_point_model_interpolates_its_points(oracle):
    src = np.array([[0, 0], [100, 0], [100, 80], [0, 80]], np.float32)
    dst = np.array([[10, 5], [120, 8], [111, 95], [2, 77]], np.float32)
    H = oracle.homography_kernel(src, dst)
    assert np.abs(project(H, src) - dst).max() < 1e-9
    H2, mask, info = oracle.find_homography(src, dst)
    assert info["found"] and mask.tolist() == [1, 1, 1, 1] and info["lm_iterations"] == 0
    assert np.array_equal(H, H2)


def test_outliers_are_rejected(oracle):
    src, dst = make_case(23, 400, 0.4, noise=0.2)
    H, mask, info = oracle.find_homography(src, dst)
    resid = np.linalg.norm(project(H_TRUE, src) - dst, axis=1)
    assert mask[resid < 1.0].all()      # true correspondences are inliers
    assert not mask[resid > 10.0].any()  # gross outliers are not
    assert info["inliers"] == int(mask.sum())
    n, m2 = oracle.homography_inliers(src, dst, H_TRUE, 3.0)
    assert n == int(m2.sum()) and abs(n - info["inliers"]) <= 3
    assert 0 < info["iterations"] < 200   # the budget collapsed: log(0.005)/log(1-0.6^4) = 39


def test_degenerate_inputs(oracle):
    few = np.zeros((3, 2), np.float32)
    H, mask, info = oracle.find_homography(few, few)
    assert H is None and not info["found"] and not mask.any()
    H, mask, info = oracle.find_homography(np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert H is None
    # collinear points: every subset is rejected, no model
    t = np.arange(20, dtype=np.float32)
    line = np.c_[t * 3, t * 2 + 1].astype(np.float32)
    H, mask, info = oracle.find_homography(line, line)
    assert H is None and info["iterations"] == 0 and not mask.any()
    # all points identical
    same = np.full((10, 2), 7, np.float32)
    H, mask, info = oracle.find_homography(same, same)
    assert H is None


def test_restatement_is_stable(oracle):
    """Regression fixture written by scripts/make_homography_fixture.py from this same restatement
    (it guards against drift; it does not pin parity with OpenCV)."""
    want = json.load(open(os.path.join(GOLDEN, "homography_cases.json")))
    for case, w in zip(CASES, want):
        src, dst = make_case(*case)
        H, mask, info = oracle.find_homography(src, dst)
        assert info == w["info"], case
        assert int(mask.sum()) == w["inliers"]
        assert [float.hex(float(v)) for v in H.ravel()] == w["H_hex"], case


# ---- HIP path against the restatement ----------------------------------------------------

def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def assert_same(ctx, oracle, src, dst, **kw):
    H, mask, info = ctx.find_homography(src, dst, **kw)
    okw = {"thresh": kw.get("reproj_threshold", 3.0), "max_iters": kw.get("max_iters", 2000),
           "confidence": kw.get("confidence", 0.995)}
    Ho, masko, infoo = oracle.find_homography(src, dst, **okw)
    assert (H is None) == (Ho is None)
    if H is not None:
        assert np.array_equal(bits(H), bits(Ho))
    assert np.array_equal(mask, masko)
    for k in infoo:
        assert info[k] == infoo[k], k
    assert info["hypotheses_scored"] >= info["iterations"] or len(src) <= 4
    return H, mask, info


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_homography_equals_restatement(ctx, oracle, case):
    src, dst = make_case(*case)
    assert_same(ctx, oracle, src, dst)


@pytest.mark.gpu
def test_gpu_homography_parameters(ctx, oracle):
    src, dst = make_case(31, 800, 0.75)
    for kw in ({"reproj_threshold": 1.0}, {"reproj_threshold": 10.0}, {"max_iters": 1}, {"max_iters": 10},
               {"max_iters": 300}, {"confidence": 0.5}, {"confidence": 0.9999}, {"max_iters": 5000, "confidence": 0.999999}):
        assert_same(ctx, oracle, src, dst, **kw)


@pytest.mark.gpu
def test_gpu_homography_degenerate_inputs(ctx, oracle, sift_lib):
    z = np.zeros((0, 2), np.float32)
    assert_same(ctx, oracle, z, z)
    assert_same(ctx, oracle, np.zeros((3, 2), np.float32), np.ones((3, 2), np.float32))
    t = np.arange(50, dtype=np.float32)
    line = np.c_[t * 3, t * 2 + 1].astype(np.float32)
    H, mask, info = assert_same(ctx, oracle, line, line)
    assert H is None
    same = np.full((10, 2), 7, np.float32)
    assert_same(ctx, oracle, same, same)
    # a grid with many collinear triples and repeated points
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8)), -1).reshape(-1, 2).astype(np.float32) * 25
    g2 = np.r_[g, g[:10]]
    assert_same(ctx, oracle, g2, (project(H_TRUE, g2)).astype(np.float32))
    with pytest.raises(sift_lib.SiftError):
        ctx.find_homography(g, g, confidence=1.0)
    with pytest.raises(sift_lib.SiftError):
        ctx.find_homography(g, g, reproj_threshold=0.0)


@pytest.mark.gpu
def test_gpu_homography_from_extracted_features(ctx, oracle, sift_lib):
    """Two overlapping crops of one synthetic scene: extract, match, keep the matches that pass the
    ratio test (main.cpp:186-198; the 3*min_dist filter of main.cpp:77-91 keeps nothing when some
    descriptors coincide exactly, as they do here), estimate: the homography is the crop offset,
    and equals the restatement's bit for bit."""
    big = sift_lib.synth_dead_leaves(5, 0, 760, 560)
    a, b = big[0:480, 0:640], big[24:504, 40:680]
    ka, da = ctx.extract(np.ascontiguousarray(a))[:2]
    kb, db = ctx.extract(np.ascontiguousarray(b))[:2]
    knn, accept = ctx.knn2(da.astype(np.uint8), db.astype(np.uint8))
    assert len(sift_lib.min_dist_filter(knn)) == 0  # min_dist is 0
    good = np.flatnonzero(accept)
    assert len(good) >= 20
    obj = np.c_[ka["x"][good], ka["y"][good]]
    scene = np.c_[kb["x"][knn["i1"][good]], kb["y"][knn["i1"][good]]]
    H, mask, info = assert_same(ctx, oracle, obj, scene)
    assert H is not None and info["inliers"] >= 20
    assert np.allclose(H, [[1, 0, -40], [0, 1, -24], [0, 0, 1]], atol=0.2)  # inliers lie within 3 px


@pytest.mark.gpu
def test_gpu_homography_with_non_finite_points(ctx, oracle):
    """NaN and infinite coordinates (a caller's bug) must neither hang nor disturb the finite correspondences:
    models drawn through them score no inliers on either side."""
    src, dst = make_case(41, 300, 0.3)
    src[5] = (np.nan, 10.0)
    dst[17] = (np.inf, -np.inf)
    src[200] = (np.inf, np.nan)
    H, mask, info = assert_same(ctx, oracle, src, dst)
    assert H is not None and np.isfinite(H).all() and info["inliers"] > 150
    assert not mask[[5, 17, 200]].any()
    bad = np.full((20, 2), np.nan, np.float32)
    H, mask, info = assert_same(ctx, oracle, bad, bad)
    assert H is None
