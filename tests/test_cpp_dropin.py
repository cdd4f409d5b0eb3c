"""The C++ drop-in layer (sift_amd/host): sift::sift_handler with the class surface
of the reference's sift.hpp, built on the C ABI."""
import os
import subprocess
import textwrap

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HOST = os.path.join(ROOT, "sift_amd", "host")


def build_host(sift_lib):
    subprocess.check_call(["make", "-C", HOST, "-s"])


def test_class_surface_compiles_like_main_cpp(sift_lib, tmp_path):
    """A translation unit that uses the handler exactly as main.cpp:36-38, 57-60 does."""
    build_host(sift_lib)
    src = tmp_path / "use.cpp"
    src.write_text(textwrap.dedent("""
        #include "sift.hpp"
        #include <type_traits>
        int main() {
            static_assert(std::is_constructible_v<sift::sift_handler, std::string, cv::Mat&&>);
            static_assert(!std::is_convertible_v<cv::Mat, sift::sift_handler>);      // explicit ctor
            static_assert(std::is_same_v<decltype(std::declval<const sift::sift_handler&>().get()), cv::Mat>);
            static_assert(std::is_same_v<decltype(sift::sift_handler::octaves), int32_t>);
            static_assert(std::is_same_v<decltype(sift::sift_handler::keypoints), std::vector<cv::KeyPoint>>);
            static_assert(std::is_same_v<decltype(sift::sift_handler::descriptors), std::vector<std::vector<double>>>);
            static_assert(std::is_same_v<decltype(sift::sift_handler::name), std::string>);
            static_assert(std::is_same_v<decltype(sift::sift_handler::base), cv::Mat>);
            static_assert(std::is_same_v<decltype(sift::sift_handler::onex), cv::Mat>);
            static_assert(sizeof(cv::KeyPoint) == 28);
            if (false) {
                cv::Mat image(8, 8, CV_8UC3);
                sift::sift_handler ss("name", std::move(image));
                ss.exec();
                auto d = ss.get();
                auto k = ss.keypoints;
                (void)d; (void)k;
                // the homography and composition calls that stand in for main.cpp:105 and :115-145
                std::vector<cv::Point2f> obj, scene;
                auto hr = sift::find_homography(obj, scene, 3);
                cv::Mat pano = sift::stitch_pair(image, image, hr.H);
                (void)pano;
            }
            auto nothing = [] {};
            TIMEIT(nothing);                                                           // sift.hpp:11-18
            return 0;
        }
    """))
    exe = tmp_path / "use"
    subprocess.check_call(["g++", "-std=c++20", "-I" + HOST, "-I" + os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L" + HOST, "-lsift_handler", "-L" + os.path.join(ROOT, "sift_amd"),
                           "-lsift_amd", "-Wl,-rpath," + HOST, "-Wl,-rpath," + os.path.join(ROOT, "sift_amd")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stderr.startswith("elapsed time: nothing: ") and out.stderr.rstrip().endswith("s")


def write_ppm(path, bgr):
    rgb = bgr[:, :, ::-1]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb).tobytes())


@pytest.mark.gpu
def test_demo_reads_ppm_like_png(sift_lib, lenna_bgr, tmp_path):
    build_host(sift_lib)
    a = tmp_path / "Lenna.ppm"
    write_ppm(a, lenna_bgr)
    out = subprocess.run([os.path.join(HOST, "sift_demo"), str(a)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Keypoints Calculated: 1112" in out.stderr


@pytest.mark.gpu
def test_demo_matches_oracle(sift_lib, oracle, lenna_bgr, lenna_rotated_bgr, golden, tmp_path):
    """BASELINE config 1 through the C++ class: Lenna vs LennaRotated detect + describe + match."""
    build_host(sift_lib)
    # the reference's own invocation, ./smai res/LennaRotated.png res/Lenna.png (PNG files read by sift_io::imread)
    a, b = os.path.join(ROOT, "tests", "golden", "LennaRotated.png"), os.path.join(ROOT, "tests", "golden", "Lenna.png")
    out = subprocess.run([os.path.join(HOST, "sift_demo"), str(a), str(b)], cwd=tmp_path, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    # stderr lines of the reference: Octaves/Size (sift.cpp:37-38), TIMEIT, counts (sift.cpp:93)
    assert "Octaves: 9" in out.stderr and "Size: [512 x 512]" in out.stderr
    assert "elapsed time: gen_gaussian_images:" in out.stderr and "elapsed time: get_descriptors:" in out.stderr
    assert "Keypoints Calculated: 1112" in out.stderr and "Keypoints Calculated: 1158" in out.stderr
    # <name>.csv (sift.cpp:114-121): x,y,size,angle at default ostream precision (6 significant digits)
    rows = np.loadtxt(tmp_path / "Lenna.csv", delimiter=",")
    ref = golden("lenna")["keypoints"]
    assert rows.shape == (1112, 4)
    for col, f in enumerate(("x", "y", "size", "angle")):
        assert np.allclose(rows[:, col], ref[f], rtol=1e-5, atol=0)
    # the pictures the reference writes beside it: "<name>_keypoints.png" (sift.cpp:89-90), "Matches<names>.png" (main.cpp:110-112)
    from PIL import Image
    assert Image.open(tmp_path / "Lenna_keypoints.png").size == (512, 512)
    assert Image.open(tmp_path / "LennaRotated_keypoints.png").size == (512, 512)
    assert Image.open(tmp_path / "MatchesLenna+LennaRotated.png").size == (1024, 512)
    # matches: query = reference image (Lenna, last argument), train = the other (main.cpp:75)
    q, t = golden("lenna")["descriptors"], golden("lennarotated")["descriptors"]
    i1, d1, i2, d2, acc = oracle.knn2_u8(q, t, threads=8)
    good = oracle.min_dist_filter(np.sqrt(d1.astype(np.float32)))
    line = [l for l in out.stdout.splitlines() if l.startswith("matches")][0].split()
    assert int(line[1]) == 1112 and int(line[3]) == len(good) and int(line[5]) == int(acc.sum())
    assert "Good Matches found:%d" % len(good) in out.stderr
    # homography over the good matches (main.cpp:92-105) equals the CPU restatement's, printed with 17 digits
    kq, kt = golden("lenna")["keypoints"], golden("lennarotated")["keypoints"]
    obj = np.c_[kq["x"][good], kq["y"][good]].astype(np.float32)
    scene = np.c_[kt["x"][i1[good]], kt["y"][i1[good]]].astype(np.float32)
    H, mask, info = oracle.find_homography(obj, scene)
    hline = [l for l in out.stdout.splitlines() if l.startswith("homography")][0].split()
    assert int(hline[2]) == info["found"] and int(hline[4]) == info["inliers"] and int(hline[6]) == info["iterations"]
    if info["found"]:
        got = np.array([[float(v) for v in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith("H ")])
        assert np.array_equal(got, H)


@pytest.mark.gpu
def test_demo_on_the_school_panorama_pair(sift_lib, oracle, school_pair, tmp_path):
    """The reference's two-image example, `smai res/school_left.jpg res/school_right.jpg`, through the C++
    class: JPEG decoding (sift_io::imread), extraction, matching, the 3*min_dist filter and the homography.
    168 good matches is the number README.md:306 publishes for it."""
    build_host(sift_lib)
    a, b = (os.path.join(ROOT, "tests", "golden", n) for n in ("school_left.jpg", "school_right.jpg"))
    out = subprocess.run([os.path.join(HOST, "sift_demo"), a, b], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Size: [800 x 600]" in out.stderr
    assert "Keypoints Calculated: 2686" in out.stderr and "Keypoints Calculated: 3472" in out.stderr
    assert "Good Matches found:168" in out.stderr
    (kl, dl), (kr, dr) = school_pair
    i1, d1, *_ = oracle.knn2_u8(dr.astype(np.uint8), dl.astype(np.uint8), threads=8)
    good = oracle.min_dist_filter(np.sqrt(d1.astype(np.float32)))
    obj = np.c_[kr["x"][good], kr["y"][good]].astype(np.float32)
    scene = np.c_[kl["x"][i1[good]], kl["y"][i1[good]]].astype(np.float32)
    H, mask, info = oracle.find_homography(obj, scene)
    hline = [l for l in out.stdout.splitlines() if l.startswith("homography")][0].split()
    assert int(hline[2]) == 1 and int(hline[4]) == info["inliers"] and int(hline[6]) == info["iterations"]
    got = np.array([[float(v) for v in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith("H ")])
    assert np.array_equal(got, H)
