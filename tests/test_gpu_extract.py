"""GPU parity tests of the extraction path: the HIP engine, called through the C
ABI, against the CPU oracle and the committed golden vectors.

Bar (stated once, used everywhere below):
  * pyramid and DoG images, candidate sets, keypoint count, and the keypoint
    fields x, y, size, response, octave: bit-exact (same IEEE operations in the
    same order on both sides);
  * angle: |diff| <= 1e-3 degrees (orientation histograms are summed in a
    different, fixed, order on the GPU and libm/ocml differ by an ulp);
  * descriptors: integer-exact for >= 99.9 % of keypoints, never off by more
    than 1 in any bin (a rounding flip of round(512*v), sift.cpp:570).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check_keypoints(kp, desc, okp, odesc):
    assert len(kp) == len(okp)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kp[f], okp[f]), f
    if len(kp):
        assert np.abs(kp["angle"].astype(np.float64) - okp["angle"]).max() <= 1e-3
        dd = np.abs(desc.astype(np.int32) - odesc.astype(np.int32))
        assert dd.max() <= 1
        assert (dd.max(axis=1) > 0).mean() <= 1e-3


def sorted_quads(c):
    return c[np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))]


def oracle_quads(o):
    c = o.candidates()
    return np.stack([c["oct"], c["img"], c["i"], c["j"]], 1).astype(np.int32)


def run_both(sift_lib, ctx, oracle, gray, check_pyramid=True):
    h, w = gray.shape
    o = oracle.OracleSift(gray).exec(threads=8, blur_threads=8)
    b = ctx.batch(w, h, 1)
    try:
        b.upload(0, gray)
        b.run(1)
        cnt = b.counts(1)[0]
        assert b.octaves == o.octaves
        if check_pyramid:
            for oc in range(o.octaves):
                assert sift_lib.octave_dims(w, h, oc) == o.level_dims(oc)
                for lvl in range(6):
                    assert np.array_equal(b.level(0, oc, lvl, sift_lib.LEVEL_GAUSS), o.gauss(oc, lvl)), (oc, lvl)
                for lvl in range(5):
                    assert np.array_equal(b.level(0, oc, lvl, sift_lib.LEVEL_DOG), o.dog(oc, lvl)), (oc, lvl)
        assert np.array_equal(sorted_quads(b.candidates(0)), oracle_quads(o))
        st = o.stats()
        assert cnt["candidates"] == int(st["candidates"].sum())
        assert cnt["oriented"] == int(st["oriented"].sum())
        kp, desc = b.download(0)
        check_keypoints(kp, desc, o.keypoints(), o.descriptors())
        return kp, desc, o
    finally:
        b.close()


def test_lenna_full_parity(sift_lib, ctx, oracle, lenna_bgr, known):
    gray = oracle.bgr_to_gray(lenna_bgr)
    kp, desc, o = run_both(sift_lib, ctx, oracle, gray)
    assert len(kp) == known["lenna"]["final_keypoints"] == 1112      # README.md:353-354


def test_lenna_rotated_full_parity(sift_lib, ctx, oracle, lenna_rotated_bgr, known):
    gray = oracle.bgr_to_gray(lenna_rotated_bgr)
    kp, _, _ = run_both(sift_lib, ctx, oracle, gray)
    assert len(kp) == known["lenna_rotated"]["final_keypoints"]


@pytest.mark.parametrize("name", ["lenna", "lennarotated", "leaves_320x240_seed7"])
def test_against_committed_golden_vectors(ctx, golden, name):
    g = golden(name)
    kp, desc, info = ctx.extract(g["gray"])
    check_keypoints(kp, desc, g["keypoints"], g["descriptors"])
    assert info["candidates"] == len(g["candidates"])


def test_bgr_entry_point_matches_gray(ctx, oracle, lenna_bgr):
    """sift_extract_bgr8 starts at cvtColor (sift.cpp:29): 15-bit integer coefficients."""
    kp_b, d_b, _ = ctx.extract(lenna_bgr)
    kp_g, d_g, _ = ctx.extract(oracle.bgr_to_gray(lenna_bgr))
    assert np.array_equal(kp_b, kp_g) and np.array_equal(d_b, d_g)


@pytest.mark.parametrize("w,h,seed", [(333, 205, 11), (130, 67, 12), (64, 257, 13), (31, 45, 14), (12, 16, 15)])
def test_ragged_sizes(sift_lib, ctx, oracle, w, h, seed):
    """Odd sizes: round-half-to-even octave sizes, partial tiles, reflect-101 borders
    wider than the image in the small octaves."""
    run_both(sift_lib, ctx, oracle, sift_lib.synth_dead_leaves(seed, 0, w, h))


def _texture(kind, seed, w, h):
    """Frames that are not piecewise constant (the dead-leaves generator's are): every pixel has a gradient, so the
    arctangent, the square root and the histogram votes see generic operands."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == "lowpass":
        k = 5
        up = np.kron(rng.integers(0, 256, (h // k + 2, w // k + 2)).astype(np.float64), np.ones((k, k)))[:h + k, :w + k]
        c = np.cumsum(np.cumsum(np.pad(up, ((1, 0), (1, 0))), 0), 1)  # k x k box filter by summed-area table
        box = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
        return np.clip(np.rint(box[:h, :w] + rng.normal(0, 2, (h, w))), 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    acc = np.zeros((h, w))
    for _ in range(12):
        th, fr, ph, am = rng.uniform(0, np.pi), rng.uniform(0.01, 0.25), rng.uniform(0, 6.28), rng.uniform(5, 30)
        acc += am * np.sin(fr * (np.cos(th) * xx + np.sin(th) * yy) + ph)
    return np.clip(np.rint(128 + acc), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind,w,h,seed", [("noise", 200, 150, 51), ("lowpass", 400, 300, 52), ("waves", 333, 517, 53)])
def test_textured_frames(sift_lib, ctx, oracle, kind, w, h, seed):
    """White noise, low-pass noise and sums of oriented sinusoids against the oracle, pyramid levels included."""
    run_both(sift_lib, ctx, oracle, _texture(kind, seed, w, h))


@pytest.mark.parametrize("w,h,seed", [(2, 2, 31), (3, 7, 32), (8, 8, 33), (5, 300, 34), (300, 5, 35), (2, 64, 36), (17, 2, 37),
                                      (40, 23, 38)])
def test_tiny_and_sliver_frames(sift_lib, ctx, oracle, w, h, seed):
    """The smallest frames the octave rule allows (sift.cpp:36) and extreme aspect ratios: every blur radius
    exceeds the image, so rows and columns are reflected several times over; pyramid and DoG must still be
    bit-identical, and nothing may be read out of bounds."""
    run_both(sift_lib, ctx, oracle, sift_lib.synth_dead_leaves(seed, 0, w, h))


@pytest.mark.parametrize("w,h,n,seed", [(333, 205, 1, 41), (415, 301, 3, 42), (97, 640, 2, 43), (1920, 1080, 1, 44), (800, 600, 12, 45)])
def test_batch_schedule_pyramid_against_oracle(sift_lib, ctx, oracle, w, h, n, seed):
    """The batch schedule (taken by the profiled pass whatever the size, and by launches of more than 4 M pixels)
    computes levels 1 and 2 of the large octaves in one launch that hands level 1 on through the LDS: strips of 48
    columns that do not divide the width, octaves too narrow for it (which fall back to one launch per level),
    one frame cut into vertical segments (1920x1080: seventeen of them) and many frames in one launch.  Every level
    of every octave of the first and the last frame must equal the oracle's."""
    frames = [sift_lib.synth_dead_leaves(seed, i, w, h) for i in range(n)]
    b = ctx.batch(w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run_profiled(n)
        for i in sorted({0, n - 1}):
            o = oracle.OracleSift(frames[i]).exec(threads=16, blur_threads=16)
            for oc in range(o.octaves):
                for lvl in range(6):
                    assert np.array_equal(b.level(i, oc, lvl, sift_lib.LEVEL_GAUSS), o.gauss(oc, lvl)), (i, oc, lvl)
            assert np.array_equal(sorted_quads(b.candidates(i)), oracle_quads(o))
            o.close()
    finally:
        b.close()


def test_natural_crop_non_multiple_of_tile(sift_lib, ctx, oracle, lenna_bgr):
    gray = oracle.bgr_to_gray(lenna_bgr)[37:37 + 301, 50:50 + 415].copy()
    run_both(sift_lib, ctx, oracle, gray)


def test_flat_image_has_no_keypoints(ctx):
    kp, desc, info = ctx.extract(np.full((96, 128), 200, np.uint8))
    assert len(kp) == 0 and desc.shape == (0, 128) and info["candidates"] == 0


def test_plateau_extrema_are_kept(sift_lib, ctx, oracle):
    """Non-strict comparisons (sift.cpp:275-276): exact plateaus qualify.  A frame
    mirrored top-bottom has exactly symmetric column sums."""
    half = sift_lib.synth_dead_leaves(21, 0, 160, 60)
    img = np.vstack([half, half[::-1]])
    run_both(sift_lib, ctx, oracle, img)


def test_batch_equals_single_and_is_deterministic(sift_lib, ctx):
    w, h, n = 320, 240, 5
    frames = [sift_lib.synth_dead_leaves(7, i, w, h) for i in range(n)]
    single = [ctx.extract(f) for f in frames]
    b = ctx.batch(w, h, n)
    for i, f in enumerate(frames):
        b.upload(i, f)
    b.run(n)
    first = [b.download(i) for i in range(n)]
    b.run(n)
    second = [b.download(i) for i in range(n)]
    b.close()
    for i in range(n):
        kp, desc = first[i]
        assert np.array_equal(kp, single[i][0])
        assert np.array_equal(desc, single[i][1].astype(np.uint8))
        assert np.array_equal(kp, second[i][0]) and np.array_equal(desc, second[i][1])


def test_batch_schedule_equals_latency_schedule(sift_lib, ctx):
    """A launch of more than 4 M pixels takes the plain launch order and the two-wave per-keypoint kernels, a single
    frame the side streams, the fused small-octave kernel and the wider per-keypoint kernels: the same frames must
    come out the same either way (the profiled pass is the plain order as well)."""
    w, h, n = 1280, 720, 6
    frames = [sift_lib.synth_dead_leaves(70, i, w, h) for i in range(n)]
    single = [ctx.extract(f) for f in frames]
    b = ctx.batch(w, h, n)
    for i, f in enumerate(frames):
        b.upload(i, f)
    b.run(n)
    plain = [b.download(i) for i in range(n)]
    b.run_profiled(n)
    profiled = [b.download(i) for i in range(n)]
    b.close()
    for i in range(n):
        assert len(single[i][0]) > 500
        for kp, desc in (plain[i], profiled[i]):
            assert np.array_equal(kp, single[i][0])
            assert np.array_equal(desc, single[i][1].astype(np.uint8))


@pytest.mark.parametrize("name", ["lenna", "lennarotated"])
def test_all_three_launch_size_tiers_agree_on_the_photographs(ctx, golden, name):
    """The per-keypoint kernels exist in three widths, chosen by the pixels of a launch (up to 1 M, up to 4 M, more):
    the orientation histogram and the descriptor tensor are summed in integers, so the variants agree by
    construction.  One 512x512 photograph run alone, eight and twenty at a time must give the same keypoints and
    descriptors, which are the oracle's."""
    g = golden(name)
    gray = g["gray"]
    h, w = gray.shape
    kp1, desc1, _ = ctx.extract(gray)
    assert len(kp1) == len(g["keypoints"])
    for n in (1, 8, 20):
        b = ctx.batch(w, h, n)
        for i in range(n):
            b.upload(i, gray)
        b.run(n)
        for i in (0, n - 1):
            kp, desc = b.download(i)
            assert np.array_equal(kp, kp1) and np.array_equal(desc, desc1.astype(np.uint8))
        b.close()


def test_keypoints_are_sorted_and_unique_like_the_reference(ctx, golden):
    kp, _, _ = ctx.extract(golden("lenna")["gray"])
    # comparator of sift.cpp:204-213 (coordinates already halved, which preserves order)
    key = list(zip(kp["x"], kp["y"], -kp["size"], kp["angle"], -kp["response"], -kp["octave"]))
    assert key == sorted(key)
    # std::unique predicate of sift.cpp:214-217 against the previous kept element
    for a, b in zip(kp[:-1], kp[1:]):
        assert abs(2 * a["x"] - 2 * b["x"]) > 1e-4 or abs(2 * a["size"] - 2 * b["size"]) > 1e-4 or \
            abs(a["angle"] - b["angle"]) > 1e-4


def test_capacity_overflow_is_reported(sift_lib, ctx, golden):
    g = golden("leaves_320x240_seed7")
    b = ctx.batch(320, 240, 1, 16)
    b.upload(0, g["gray"])
    b.run(1)
    with pytest.raises(sift_lib.SiftError) as e:
        b.counts(1)
    assert e.value.code == sift_lib.api.SIFT_ERR_OVERFLOW
    b.close()


def test_get_returns_float_copy(sift_lib, ctx, golden):
    """get() (sift.cpp:103-112) narrows the descriptors to float32."""
    import ctypes as C
    g = golden("leaves_320x240_seed7")
    L = sift_lib.lib()
    res = sift_lib.api._Result()
    gray = np.ascontiguousarray(g["gray"])
    assert L.sift_extract_gray8(ctx._h, gray.ctypes.data, 320, 240, 320, C.byref(res)) == 0
    out = np.zeros((res.n, 128), np.float32)
    assert L.sift_result_get_f32(C.byref(res), out.ctypes.data) == 0
    L.sift_result_free(C.byref(res))
    assert np.array_equal(out, g["descriptors"].astype(np.float32))


def test_1080p_frame_against_oracle(sift_lib, ctx, oracle, known):
    """BASELINE config 3 frame 0 at full size: candidates, keypoints and descriptors."""
    img = sift_lib.synth_dead_leaves(1, 0, 1920, 1080)
    kp, desc, o = run_both(sift_lib, ctx, oracle, img, check_pyramid=False)
    k = known["dead_leaves_1920x1080_seed1_frame0"]
    assert len(kp) == k["final_keypoints"]
    assert int(o.stats()["candidates"].sum()) == k["candidates"]


def test_1080p_pyramid_properties(sift_lib, ctx):
    """Size-independent properties at full size: D[j-1] == G[j] - G[j-1], next octave
    == even pixels of level 3, blur preserves the mean of a constant-extended image."""
    img = sift_lib.synth_dead_leaves(1, 3, 1920, 1080)
    b = ctx.batch(1920, 1080, 1)
    b.upload(0, img)
    b.run(1)
    b.counts(1)
    for oc in (0, 1, 4, 9):
        g = [b.level(0, oc, l, sift_lib.LEVEL_GAUSS) for l in range(6)]
        for l in range(5):
            assert np.array_equal(b.level(0, oc, l, sift_lib.LEVEL_DOG), g[l + 1] - g[l])
        if oc + 1 < b.octaves:
            nxt = b.level(0, oc + 1, 0, sift_lib.LEVEL_GAUSS)
            assert np.array_equal(nxt, g[3][::2, ::2][: nxt.shape[0], : nxt.shape[1]])
        assert 0 <= g[5].min() and g[5].max() <= 255.0000001
    b.close()


def test_4k_frame_runs_and_matches_oracle_counts(sift_lib, ctx, oracle):
    """BASELINE config 4 (3840x2160, 11 octaves, 7680x4320 base)."""
    img = sift_lib.synth_dead_leaves(3, 0, 3840, 2160)
    kp, desc, info = ctx.extract(img)
    assert info["octaves"] == 11
    o = oracle.OracleSift(img).exec(threads=16, blur_threads=16)
    check_keypoints(kp, desc.astype(np.uint8), o.keypoints(), o.descriptors())


@pytest.mark.gpu
def test_single_image_call_grows_its_workspace(ctx, oracle, lenna_bgr):
    """The reference's vectors grow without bound; the single-image call runs a frame that exceeds its
    workspace again in a larger one.  Forced here by a tiny initial capacity."""
    gray = oracle.bgr_to_gray(lenna_bgr)
    o = oracle.OracleSift(gray).exec(threads=8, blur_threads=8)
    try:
        ctx.set_single_image_capacity(100)
        kps, desc, info = ctx.extract(gray)
        assert info["candidates"] == int(o.stats()["candidates"].sum()) > 100
        check_keypoints(kps, desc, o.keypoints(), o.descriptors())
        kps2, desc2, _ = ctx.extract(gray)  # the grown workspace is reused
        assert np.array_equal(kps, kps2) and np.array_equal(desc, desc2)
    finally:
        ctx.set_single_image_capacity(0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["school_left.jpg", "school_right.jpg"])
def test_natural_800x600_photographs(sift_lib, ctx, oracle, name):
    """The reference's panorama example frames (res/school_*.jpg): natural content, large flat and saturated areas."""
    from conftest import _load_rgb
    gray = oracle.bgr_to_gray(_load_rgb(name)[:, :, ::-1].copy())
    kp, desc, o = run_both(sift_lib, ctx, oracle, gray, check_pyramid=False)
    assert len(kp) == {"school_left.jpg": 2686, "school_right.jpg": 3472}[name]
    assert np.array_equal(desc, o.descriptors().astype(desc.dtype))
    assert np.array_equal(kp["angle"], o.keypoints()["angle"])


def _patterns():
    yy, xx = np.mgrid[0:150, 0:210]
    rng = np.random.default_rng(3)
    return {
        "checkerboard8": (((yy // 8 + xx // 8) & 1) * 255).astype(np.uint8),
        "checkerboard1": (((yy + xx) & 1) * 255).astype(np.uint8),
        "stripes": ((xx // 5 & 1) * 200 + 20).astype(np.uint8),
        "one_pixel": np.where((yy == 70) & (xx == 100), 255, 0).astype(np.uint8),
        "white": np.full((150, 210), 255, np.uint8),
        "ramp": ((xx + yy) % 256).astype(np.uint8),
        "binary_noise": (rng.integers(0, 2, (150, 210)) * 255).astype(np.uint8),
        "discs": np.where(((yy - 75) ** 2 + (xx - 105) ** 2 < 40 ** 2) ^ ((yy - 60) ** 2 + (xx - 90) ** 2 < 12 ** 2), 230, 25).astype(np.uint8),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_patterns()))
def test_regular_and_degenerate_patterns(sift_lib, ctx, oracle, name):
    """Periodic patterns produce exactly repeated values (ties in the non-strict extremum test and in the sort
    keys), saturated and constant frames produce none: everything must still agree with the oracle."""
    run_both(sift_lib, ctx, oracle, _patterns()[name])


@pytest.mark.gpu
def test_repeated_small_runs_are_identical(sift_lib, ctx):
    """Small workloads are scheduled on two streams; 150 back-to-back runs of frames of different sizes must
    give the same keypoints and descriptors every time (a missing dependency would show as a flicker)."""
    frames = [sift_lib.synth_dead_leaves(50 + k, 0, w, h) for k, (w, h) in enumerate(((320, 240), (512, 512), (97, 131)))]
    want = [ctx.extract(f)[:2] for f in frames]
    for it in range(150):
        k = it % len(frames)
        kp, desc, _ = ctx.extract(frames[k])
        assert np.array_equal(kp, want[k][0]) and np.array_equal(desc, want[k][1]), (it, k)
    b = ctx.batch(320, 240, 3)
    try:
        for i in range(3):
            b.upload(i, frames[0])
        for it in range(50):
            b.run(3)
            b.counts(3)
            for i in range(3):
                kp, desc = b.download(i)
                assert np.array_equal(kp, want[0][0]) and np.array_equal(desc.astype(np.float64), want[0][1]), (it, i)
    finally:
        b.close()


@pytest.mark.gpu
def test_base_image_is_the_constructors_base(ctx, oracle, golden):
    """sift_base_image_gray8 = resize x2 + GaussianBlur of the constructor (sift.cpp:40-45), bit for bit."""
    gray = golden("leaves_320x240_seed7")["gray"]
    o = oracle.OracleSift(gray).exec(threads=4)
    base = ctx.base_image(gray)
    assert base.shape == (480, 640) and np.array_equal(base, o.gauss(0, 0))
    kp, desc, _ = ctx.extract(gray)                 # the workspace it shares with extract() is left usable
    g = golden("leaves_320x240_seed7")
    check_keypoints(kp, desc, g["keypoints"], g["descriptors"])


@pytest.mark.gpu
def test_stage_timed_call_returns_the_same_result_and_five_times(ctx, golden):
    g = golden("lenna")
    kp, desc, _ = ctx.extract(g["gray"])
    kp2, desc2, ms = ctx.extract_stages(g["gray"])
    assert np.array_equal(kp, kp2) and np.array_equal(desc, desc2)
    assert len(ms) == 5 and ms[1] == 0.0 and all(m >= 0 for m in ms) and 0.05 < sum(ms) < 100
    assert ms[0] > 0 and ms[2] > 0 and ms[4] > 0


@pytest.mark.gpu
def test_2048x1080_frame_standing_in_for_2k_png(sift_lib, ctx, oracle):
    """BASELINE config 2 names res/2k.png, which the reference checkout does not contain; one dead-leaves frame of
    2048 x 1080 (seed 2) stands in for it: full pipeline against the oracle."""
    img = sift_lib.synth_dead_leaves(2, 0, 2048, 1080)
    kp, desc, o = run_both(sift_lib, ctx, oracle, img, check_pyramid=False)
    assert len(kp) > 3000
    assert np.array_equal(desc, o.descriptors().astype(desc.dtype))


def _numpy_gradients(L):
    """sift.cpp:410-415 per pixel in doubles, with the C library's atan2 (math.atan2), which is what the reference
    and the oracle call; numpy's own vectorised arctan2 is a few ulps less accurate and decides exact-diagonal
    gradients differently."""
    import math
    dx = np.zeros_like(L)
    dy = np.zeros_like(L)
    dx[1:-1, 1:-1] = L[1:-1, 2:] - L[1:-1, :-2]
    dy[1:-1, 1:-1] = L[:-2, 1:-1] - L[2:, 1:-1]
    mag = np.sqrt(dx * dx + dy * dy)
    ang = (np.frompyfunc(math.atan2, 2, 1)(dy, dx).astype(np.float64) * 180) / np.pi
    inner = np.zeros(L.shape, bool)
    inner[1:-1, 1:-1] = True
    return np.where(inner, mag, 0.0), np.where(inner, ang, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["leaves", "diagonal_symmetry", "lenna", "tiny_and_huge"])
def test_gradient_planes_against_libm(sift_lib, ctx, golden, kind):
    """The device evaluates atan2 with its own table-and-series routine (hardware reciprocal seeds) and refines
    results that land next to one of the 36 histogram bin boundaries; host libm is the reference's arithmetic.
    Gradient magnitudes must be bit-identical, directions within 4 ulps of the degree value (two ulps of the
    arctangent in radians, carried through the *180/pi conversion), and the orientation-histogram bin
    round(direction*36/360) (sift.cpp:416) identical for every pixel of every plane."""
    if kind == "leaves":
        img = sift_lib.synth_dead_leaves(9, 0, 333, 205)
    elif kind == "diagonal_symmetry":           # img == img.T: gradients exactly on the 45-degree directions
        a = sift_lib.synth_dead_leaves(10, 0, 200, 200).astype(np.int32)
        img = ((a + a.T) // 2).astype(np.uint8)
    elif kind == "lenna":
        img = golden("lenna")["gray"]
    else:                                        # flat areas with single-level steps and full-range edges
        yy, xx = np.mgrid[0:160, 0:224]
        img = np.where((xx // 16 + yy // 16) % 2 == 0, 100, 101).astype(np.uint8)
        img[40:120, 60:180] = np.where((xx[40:120, 60:180] * 3 + yy[40:120, 60:180]) % 7 < 3, 0, 255)
    h, w = img.shape
    b = ctx.batch(w, h, 1)
    try:
        b.upload(0, img)
        b.run(1)
        b.counts(1)
        checked = 0
        for oc in range(b.octaves):
            ow, oh = sift_lib.octave_dims(w, h, oc)
            if ow <= 10 or oh <= 10:
                with pytest.raises(sift_lib.SiftError):
                    b.level(0, oc, 1, sift_lib.LEVEL_GRAD_MAG)
                continue
            for lvl in (1, 2, 3):
                L = b.level(0, oc, lvl, sift_lib.LEVEL_GAUSS)
                mag, ang = _numpy_gradients(L)
                gm = b.level(0, oc, lvl, sift_lib.LEVEL_GRAD_MAG)
                gd = b.level(0, oc, lvl, sift_lib.LEVEL_GRAD_DIR)
                assert np.array_equal(gm, mag), (oc, lvl)
                err = np.abs(gd - ang)
                assert (err <= 4 * np.spacing(np.maximum(np.abs(ang), 1e-300))).all(), (oc, lvl, err.max())
                bins = lambda a: np.sign(a * 36 / 360) * np.floor(np.abs(a * 36 / 360) + 0.5)   # std::round
                bad = np.argwhere(bins(gd) != bins(ang))
                assert len(bad) == 0, (oc, lvl, len(bad), [(int(y), int(x), float(gd[y, x]).hex(), float(ang[y, x]).hex()) for y, x in bad[:4]])
                checked += gm.size
        assert checked > 10000
    finally:
        b.close()
