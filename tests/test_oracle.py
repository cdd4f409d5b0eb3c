"""CPU tests of the oracle: it is pinned against the reference's published known
answer (1112 keypoints on the 512x512 Lenna, README.md:353-354) and the stage
checkpoints in tests/golden/known_answers.json, and must reproduce the
committed golden vectors."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def lenna(oracle, lenna_bgr):
    gray = oracle.bgr_to_gray(lenna_bgr)
    o = oracle.OracleSift(gray).exec(threads=8)
    return gray, o


def test_gray_conversion_known_answer(lenna, known):
    gray, _ = lenna
    k = known["lenna"]
    assert gray[0, :4].tolist() == k["gray_first4"]
    assert int(gray.sum(dtype=np.int64)) == k["gray_sum"]


def test_gaussian_kernel_known_answer(oracle, known):
    k = known["lenna"]
    taps = oracle.gaussian_kernel(np.sqrt(1.6 ** 2 - 4 * 0.5 ** 2))
    assert len(taps) == k["base_kernel_taps"]
    assert taps[0] == pytest.approx(k["base_kernel_k0"], rel=1e-14)
    assert taps[5] == pytest.approx(k["base_kernel_k5"], rel=1e-14)
    assert np.array_equal(taps, taps[::-1])
    # incremental sigmas of sift.cpp:134-141 give 11, 13, 17, 21, 27 taps
    kk, prev, n = 2 ** (1 / 3), 1.6, []
    for _ in range(5):
        now = prev * kk
        n.append(len(oracle.gaussian_kernel(np.sqrt(now * now - prev * prev))))
        prev = now
    assert n == [11, 13, 17, 21, 27]


def test_pyramid_checkpoints(lenna, known):
    _, o = lenna
    k = known["lenna"]
    assert o.octaves == k["octaves"]
    assert [list(o.level_dims(i)) for i in range(o.octaves)] == k["octave_sizes"]
    base = o.base()
    assert base.sum() == pytest.approx(k["base_sum"], abs=1e-5)
    assert base[0, 0] == pytest.approx(k["base_0_0"], abs=1e-11)
    assert base[512, 512] == pytest.approx(k["base_512_512"], abs=1e-11)
    assert np.array_equal(o.gauss(0, 0), base)
    for oct_, lvl, y, x, v in k["gauss"]:
        assert o.gauss(oct_, lvl)[y, x] == pytest.approx(v, abs=1e-11)
    for oct_, lvl, y, x, v in k["dog"]:
        assert o.dog(oct_, lvl)[y, x] == pytest.approx(v, abs=1e-11)
    for lvl, v in enumerate(k["dog0_abs_sums"]):
        assert np.abs(o.dog(0, lvl)).sum() == pytest.approx(v, abs=2e-3)
    # gen_dog_images: D[j-1] = G[j] - G[j-1] (sift.cpp:173)
    assert np.array_equal(o.dog(1, 2), o.gauss(1, 3) - o.gauss(1, 2))
    # next octave = nearest-neighbour half of level 3 (sift.cpp:152-153)
    assert np.array_equal(o.gauss(1, 0), o.gauss(0, 3)[::2, ::2])


def test_extrema_counts_known_answer(lenna, known):
    _, o = lenna
    k = known["lenna"]
    st = o.stats()
    assert int(st["candidates"].sum()) == k["candidates"]
    assert int(st["no_convergence"].sum()) == k["no_convergence"]
    assert int(st["low_contrast"].sum()) == k["low_contrast"]
    assert int(st["det_nonpositive"].sum()) == k["det_nonpositive"]
    assert int(st["edge_like"].sum()) == k["edge_like"]
    assert int(st["out_of_bounds"].sum() + st["singular"].sum()) == k["out_of_bounds_plus_singular"]
    assert int(st["localized"].sum()) == k["localized"]
    assert int(st["oriented"].sum()) == k["oriented"]
    for oct_, img, nc, no in k["per_layer_candidates_oriented"]:
        assert int(st[oct_, img - 1]["candidates"]) == nc
        assert int(st[oct_, img - 1]["oriented"]) == no
    c = o.candidates()
    first = c[(c["oct"] == 0) & (c["img"] == 1)][:5]
    assert [[int(r["i"]), int(r["j"])] for r in first] == k["first_candidates_oct0_img1"]


def test_published_keypoint_count(lenna, known):
    """README.md:353-354: the 512x512 image yields 1112 keypoints."""
    _, o = lenna
    k = known["lenna"]
    kp = o.keypoints()
    assert len(kp) == k["final_keypoints"] == 1112
    # first sorted keypoints (recorded before the halving of sift.cpp:219-223)
    for row, i in zip(k["first_sorted_prehalving"], (0, 2)):
        assert 2 * kp[i]["x"] == pytest.approx(row[0], rel=1e-6)
        assert 2 * kp[i]["y"] == pytest.approx(row[1], rel=1e-6)
        assert 2 * kp[i]["size"] == pytest.approx(row[2], rel=1e-6)
        assert kp[i]["angle"] == pytest.approx(row[3], rel=1e-6)
        assert kp[i]["response"] == pytest.approx(row[4], rel=1e-6)
        assert (kp[i]["octave"] & ~255) | ((kp[i]["octave"] + 1) & 255) == row[5]
    assert int((kp["angle"] > 360).sum()) == k["angles_above_360"]
    assert kp["angle"].min() == pytest.approx(k["angle_min"], abs=1e-2)
    assert kp["angle"].max() == pytest.approx(k["angle_max"], abs=1e-2)
    d = o.descriptors()
    assert d.shape == (1112, 128) and d.min() >= 0 and d.max() <= 102
    assert np.array_equal(d, np.round(d))
    assert np.array_equal(o.get(), d.astype(np.float32))


def test_lenna_rotated_counts(oracle, lenna_rotated_bgr, known):
    o = oracle.OracleSift(oracle.bgr_to_gray(lenna_rotated_bgr)).exec(threads=8)
    k = known["lenna_rotated"]
    st = o.stats()
    assert int(st["candidates"].sum()) == k["candidates"]
    assert int(st["localized"].sum()) == k["localized"]
    assert int(st["oriented"].sum()) == k["oriented"]
    assert len(o.keypoints()) == k["final_keypoints"]


def test_dead_leaves_960x540_counts(oracle, sift_lib, known):
    img = sift_lib.synth_dead_leaves(1, 0, 960, 540)
    o = oracle.OracleSift(img).exec(threads=8)
    k = known["dead_leaves_960x540_seed1_frame0"]
    st = o.stats()
    assert int(st["candidates"].sum()) == k["candidates"]
    assert int(st["localized"].sum()) == k["localized"]
    assert int(st["oriented"].sum()) == k["oriented"]
    assert len(o.keypoints()) == k["final_keypoints"]
    # the survey's independent numpy restatement agrees on everything past the
    # tie-sensitive candidate count (see _dead_leaves_note in known_answers.json)
    assert k["localized"] == k["probe_localized"] and k["oriented"] == k["probe_oriented"]
    assert k["final_keypoints"] == k["probe_final_keypoints"]


@pytest.mark.parametrize("name", ["lenna", "lennarotated", "leaves_320x240_seed7"])
def test_oracle_reproduces_golden_vectors(oracle, golden, name):
    g = golden(name)
    o = oracle.OracleSift(g["gray"]).exec(threads=4)
    assert np.array_equal(o.keypoints(), g["keypoints"])
    assert np.array_equal(o.descriptors().astype(np.uint8), g["descriptors"])
    c = o.candidates()
    assert np.array_equal(np.stack([c["oct"], c["img"], c["i"], c["j"]], 1), g["candidates"])


def test_thread_count_does_not_change_results(oracle, golden):
    g = golden("leaves_320x240_seed7")
    a = oracle.OracleSift(g["gray"]).exec(threads=1, blur_threads=1)
    b = oracle.OracleSift(g["gray"]).exec(threads=7, blur_threads=3)
    assert np.array_equal(a.keypoints(), b.keypoints())
    assert np.array_equal(a.descriptors(), b.descriptors())


def test_unique_predicate_ignores_y(oracle):
    """Quirk of sift.cpp:215-216: pt.x is compared twice and pt.y never, so two
    keypoints that differ only in y are 'duplicates'.  A flat image has no
    keypoints at all (empty result is legal)."""
    o = oracle.OracleSift(np.full((64, 64), 77, np.uint8)).exec(threads=2)
    assert len(o.keypoints()) == 0
    assert o.descriptors().shape == (0, 128)


def test_knn_oracle_small_cases(oracle):
    q = np.array([[0] * 128, [5] * 128], np.uint8)
    t = np.array([[1] * 128, [0] * 128, [1] * 128, [9] * 128], np.uint8)
    i1, d1, i2, d2, acc = oracle.knn2_u8(q, t)
    assert i1.tolist() == [1, 0] and d1.tolist() == [0, 16 * 128]    # q1: rows 0, 2, 3 tie at 2048
    assert i2.tolist() == [0, 2] and d2.tolist() == [128, 16 * 128]  # ties -> smaller index first
    assert acc.tolist() == [1, 0]
    fi1, fd1, fi2, fd2, facc = oracle.knn2_f32(q.astype(np.float32), t.astype(np.float32))
    assert fi1.tolist() == i1.tolist() and fi2.tolist() == i2.tolist() and facc.tolist() == acc.tolist()
    assert fd1.tolist() == d1.tolist() and fd2.tolist() == d2.tolist()
    # one train row: no second neighbour, nothing accepted
    i1, d1, i2, d2, acc = oracle.knn2_u8(q, t[:1])
    assert i1.tolist() == [0, 0] and i2.tolist() == [-1, -1] and acc.tolist() == [0, 0]


def test_min_dist_filter_oracle(oracle):
    dist = np.array([10.0, 31.0, 29.9, 500.0, 3.0, 8.9, 9.0], np.float32)
    assert oracle.min_dist_filter(dist).tolist() == [4, 5]       # min 3 -> good < 9
    assert oracle.min_dist_filter(np.array([400.0, 250.0], np.float32)).tolist() == [1]  # min_dist stays 100


def test_school_pair_gives_the_published_168_good_matches(oracle, school_pair):
    """README.md:306 (and docs/Final Report.md:306) caption the two-image panorama example with
    "Feature matching (168 good matches)".  `smai school_left.jpg school_right.jpg` takes the last image as
    the query (main.cpp:41, :75) and keeps matches with distance < 3 * min_dist (main.cpp:77-91): the oracle's
    keypoints and descriptors, matched exactly, give the same 168 (the other argument order gives 167)."""
    (kl, dl), (kr, dr) = school_pair
    assert (len(kl), len(kr)) == (2686, 3472)
    i1, d1, i2, d2, acc = oracle.knn2_u8(dr.astype(np.uint8), dl.astype(np.uint8), threads=8)
    good = oracle.min_dist_filter(np.sqrt(d1.astype(np.float32)))
    assert len(good) == 168
    i1r, d1r, *_ = oracle.knn2_u8(dl.astype(np.uint8), dr.astype(np.uint8), threads=8)
    assert len(oracle.min_dist_filter(np.sqrt(d1r.astype(np.float32)))) == 167
    # the matches describe one homography: RANSAC keeps nearly all of them
    obj = np.c_[kr["x"][good], kr["y"][good]]
    scene = np.c_[kl["x"][i1[good]], kl["y"][i1[good]]]
    H, mask, info = oracle.find_homography(obj, scene)
    assert info["found"] and info["inliers"] >= 160


def test_blur_example_does_not_reproduce_the_readmes_eight(oracle, lenna_bgr):
    """README.md:425 says of its blurring example "Number of keypoints matched are reduced to 8" without naming the
    image pair or the matcher settings (the figure was made with the approximate FLANN matcher).  The natural
    candidate, res/Lenna_blur.png (a data file of the reference, copied to tests/golden) against res/Lenna.png
    through the live 3*min_dist filter (main.cpp:77-91), gives 11 good matches in either argument order with this
    oracle, not 8, so that number is NOT a pin; what the oracle gives is recorded here so that a change shows."""
    from conftest import _load_rgb
    blur = _load_rgb("Lenna_blur.png")[:, :, ::-1].copy()
    ob = oracle.OracleSift(oracle.bgr_to_gray(blur)).exec(threads=8, blur_threads=8)
    ol = oracle.OracleSift(oracle.bgr_to_gray(lenna_bgr)).exec(threads=8, blur_threads=8)
    db, dl = ob.descriptors().astype(np.uint8), ol.descriptors().astype(np.uint8)
    assert (len(dl), len(db)) == (1112, 30)
    for q, t in ((dl, db), (db, dl)):
        i1, d1, *_ = oracle.knn2_u8(q, t, threads=8)
        assert len(oracle.min_dist_filter(np.sqrt(d1.astype(np.float32)))) == 11
