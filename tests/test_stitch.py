"""Panorama composition, main.cpp:115-145: warpPerspective, hole filling, 0.7/0.3 blend, crop.

PARITY WITH OPENCV IS UNPINNED (OpenCV is not installed and the reference holds no stitched output): the CPU
restatement (oracle/stitch_oracle.c) is checked against ground truth and independent arithmetic, the HIP path
against the restatement bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import _load_rgb

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HOST = os.path.join(ROOT, "sift_amd", "host")

# the homography the school pair's 168 good matches give (tests/test_cpp_dropin.py checks its bits elsewhere)
H_SCHOOL = np.array([[7.39989182e-01, 3.48067351e-02, 5.01099764e+02],
                     [-1.20732099e-01, 9.50325866e-01, 4.69853618e+00],
                     [-3.18268396e-04, 2.77490327e-07, 1.0]])

HOMOGRAPHIES = {
    "identity": np.eye(3),
    "shift": np.array([[1, 0, 10.5], [0, 1, -3.25], [0, 0, 1.0]]),
    "rotate_scale": np.array([[0.9, -0.35, 40.0], [0.35, 0.9, -25.0], [0, 0, 1.0]]),
    "perspective": np.array([[1.1, 0.05, 30.0], [-0.03, 0.95, -12.0], [2e-4, -1e-4, 1.0]]),
    "far_away": np.array([[1, 0, 5000.0], [0, 1, 7000.0], [0, 0, 1.0]]),
    "horizon": np.array([[1, 0.2, 3.0], [0.1, 1, 2.0], [4e-3, 1e-3, 1.0]]),   # w crosses zero inside the canvas
    "school": H_SCHOOL,
}


@pytest.fixture(scope="module")
def photo():
    return np.ascontiguousarray(_load_rgb("school_left.jpg")[:, :, ::-1][100:340, 200:520])


def float_bilinear(src, H, dw, dh):
    Minv = np.linalg.inv(H)
    yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
    w = Minv[2, 0] * xx + Minv[2, 1] * yy + Minv[2, 2]
    sx = (Minv[0, 0] * xx + Minv[0, 1] * yy + Minv[0, 2]) / w
    sy = (Minv[1, 0] * xx + Minv[1, 1] * yy + Minv[1, 2]) / w
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0

    def tap(y, x):
        ok = (x >= 0) & (x < src.shape[1]) & (y >= 0) & (y < src.shape[0])
        out = np.zeros((dh, dw, 3))
        out[ok] = src[y[ok], x[ok]]
        return out
    return (tap(y0, x0) * ((1 - fx) * (1 - fy))[..., None] + tap(y0, x0 + 1) * (fx * (1 - fy))[..., None] +
            tap(y0 + 1, x0) * ((1 - fx) * fy)[..., None] + tap(y0 + 1, x0 + 1) * (fx * fy)[..., None])


# ---- the restatement (CPU) ---------------------------------------------------------------

def test_identity_warp_is_exact_and_padding_is_black(oracle, photo):
    h, w = photo.shape[:2]
    out = oracle.warp_perspective(photo, np.eye(3), w + 17, h + 5)
    assert np.array_equal(out[:h, :w], photo)
    assert not out[h:].any() and not out[:, w:].any()


@pytest.mark.parametrize("name", ["shift", "rotate_scale", "perspective", "school"])
def test_warp_is_bilinear_interpolation(oracle, photo, name):
    """Coordinates are rounded to 1/32 pixel and weights to 15 bits, so the result stays within a few grey
    levels of exact bilinear interpolation where the picture is smooth, and mostly within one."""
    H = HOMOGRAPHIES[name]
    dw, dh = 400, 300
    out = oracle.warp_perspective(photo, H, dw, dh).astype(np.float64)
    ref = float_bilinear(photo, H, dw, dh)
    d = np.abs(out - ref)
    assert np.median(d) <= 0.5 and np.mean(d > 4) < 0.02


def test_singular_homography_and_tall_image_are_refused(oracle, photo):
    assert oracle.warp_perspective(photo, np.zeros((3, 3)), 10, 10) is None
    tall = np.zeros((photo.shape[0] + 1, 8, 3), np.uint8)
    assert oracle.stitch_pair(photo, tall, np.eye(3)) is None


def test_compose_rules(oracle):
    """main.cpp:120-145 on a case small enough to do by hand."""
    ref = np.zeros((4, 3, 3), np.uint8)
    ref[:, :, :] = (10, 0, 200)              # the green channel is zero everywhere: it is a hole per channel
    img = np.full((2, 2, 3), 100, np.uint8)
    out = oracle.stitch_pair(ref, img, np.eye(3))
    # canvas 5 columns wide; the last two columns stay black and are cropped
    assert out.shape == (4, 3, 3)
    # inside img's rectangle: blue 0.7*100+0.3*10 = 73, green hole -> 100, red 0.7*100+0.3*200 = 130
    assert out[0, 0].tolist() == [73, 100, 130] and out[1, 1].tolist() == [73, 100, 130]
    assert out[3, 2].tolist() == [10, 0, 200] and out[0, 2].tolist() == [10, 0, 200]


def test_school_panorama(oracle):
    right = _load_rgb("school_right.jpg")[:, :, ::-1].copy()
    left = _load_rgb("school_left.jpg")[:, :, ::-1].copy()
    pano = oracle.stitch_pair(right, left, H_SCHOOL)
    assert pano.shape == (600, 1493, 3)
    # left image region is a blend dominated by the left image; the part beyond it is the warped right image
    assert np.abs(pano[:, :800].astype(int) - left.astype(int)).mean() < 12
    assert pano[:, 900:1400].mean() > 60


# ---- the HIP path against the restatement ----------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(HOMOGRAPHIES))
def test_gpu_warp_equals_restatement(ctx, oracle, photo, name):
    H = HOMOGRAPHIES[name]
    for dw, dh in ((400, 300), (63, 15), (65, 9), (1, 1), (130, 16)):   # short canvases change the block width
        got = ctx.warp_perspective(photo, H, dw, dh)
        assert np.array_equal(got, oracle.warp_perspective(photo, H, dw, dh)), (name, dw, dh)


@pytest.mark.gpu
def test_gpu_stitch_equals_restatement(ctx, oracle, sift_lib):
    right = _load_rgb("school_right.jpg")[:, :, ::-1].copy()
    left = _load_rgb("school_left.jpg")[:, :, ::-1].copy()
    got = ctx.stitch_pair(right, left, H_SCHOOL)
    assert np.array_equal(got, oracle.stitch_pair(right, left, H_SCHOOL))
    # a strided view as input, a second panorama on top of the first (main.cpp:151)
    got2 = ctx.stitch_pair(got, left[:500, 10:700], HOMOGRAPHIES["shift"])
    assert np.array_equal(got2, oracle.stitch_pair(got, np.ascontiguousarray(left[:500, 10:700]), HOMOGRAPHIES["shift"]))
    with pytest.raises(sift_lib.SiftError):
        ctx.stitch_pair(left[:100], left, np.eye(3))       # new image taller than the panorama
    with pytest.raises(sift_lib.SiftError):
        ctx.warp_perspective(left, np.zeros((3, 3)), 8, 8)  # singular


@pytest.mark.gpu
def test_demo_writes_the_school_panorama(sift_lib, oracle, tmp_path):
    """`smai res/school_left.jpg res/school_right.jpg` to the end: Final-school_right+school_left.png holds the
    restatement's panorama for the homography the run printed."""
    from PIL import Image
    import test_cpp_dropin
    test_cpp_dropin.build_host(sift_lib)
    a, b = (os.path.join(ROOT, "tests", "golden", n) for n in ("school_left.jpg", "school_right.jpg"))
    out = subprocess.run([os.path.join(HOST, "sift_demo"), a, b], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    H = np.array([[float(v) for v in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith("H ")])
    pano = np.array(Image.open(tmp_path / "Final-school_right+school_left.png").convert("RGB"))[:, :, ::-1]
    want = oracle.stitch_pair(_load_rgb("school_right.jpg")[:, :, ::-1].copy(), _load_rgb("school_left.jpg")[:, :, ::-1].copy(), H)
    assert np.array_equal(pano, want)
    assert "[%d x %d]" % (want.shape[1], want.shape[0]) in out.stderr


def oracle_main(oracle, images, names):
    """The reference's main.cpp:41-152 with the CPU oracle: returns {file name: panorama}."""
    ref, name_ref = images[-1], names[-1]
    out = {}
    for k in range(len(images) - 2, -1, -1):
        o_ref = oracle.OracleSift(oracle.bgr_to_gray(ref)).exec(threads=8, blur_threads=8)
        o_img = oracle.OracleSift(oracle.bgr_to_gray(images[k])).exec(threads=8, blur_threads=8)
        kq, dq, kt, dt = o_ref.keypoints(), o_ref.descriptors(), o_img.keypoints(), o_img.descriptors()
        i1, d1, *_ = oracle.knn2_u8(dq.astype(np.uint8), dt.astype(np.uint8), threads=8)
        good = oracle.min_dist_filter(np.sqrt(d1.astype(np.float32)))
        obj = np.c_[kq["x"][good], kq["y"][good]].astype(np.float32)
        scene = np.c_[kt["x"][i1[good]], kt["y"][i1[good]]].astype(np.float32)
        H, mask, info = oracle.find_homography(obj, scene)
        name_ref += "+" + names[k]
        assert H is not None, name_ref
        ref = oracle.stitch_pair(ref, images[k], H)
        out["Final-%s.png" % name_ref] = ref
    return out


@pytest.mark.gpu
def test_demo_chains_three_images_like_main(sift_lib, oracle, tmp_path):
    """Three overlapping views of one synthetic scene, each with its own sensor noise, given as
    `smai a.png b.png c.png`: the panorama grows from the last image backwards (main.cpp:54-151) and every
    Final-*.png equals what the CPU restatements of all stages produce."""
    from PIL import Image
    import test_cpp_dropin
    test_cpp_dropin.build_host(sift_lib)
    scene = sift_lib.synth_dead_leaves(77, 0, 900, 360)
    rng = np.random.default_rng(5)
    names, images = ["a", "b", "c"], []
    for n, x0 in zip(names, (0, 190, 380)):
        view = scene[:, x0:x0 + 420].astype(np.int16) + rng.integers(-3, 4, (360, 420))
        g = np.clip(view, 0, 255).astype(np.uint8)
        bgr = np.stack([g, g, g], axis=2)
        images.append(bgr)
        Image.fromarray(bgr).save(tmp_path / (n + ".png"))
    want = oracle_main(oracle, images, names)
    out = subprocess.run([os.path.join(HOST, "sift_demo")] + [str(tmp_path / (n + ".png")) for n in names], cwd=tmp_path,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert sorted(want) == ["Final-c+b+a.png", "Final-c+b.png"]
    for fname, pano in want.items():
        got = np.array(Image.open(tmp_path / fname).convert("RGB"))[:, :, ::-1]
        assert got.shape == pano.shape and np.array_equal(got, pano), fname
    assert want["Final-c+b+a.png"].shape[1] > want["Final-c+b.png"].shape[1] > 420
