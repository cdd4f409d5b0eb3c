import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built with gcc if the .so is absent."""
    from oracle import oracle as O
    if not os.path.exists(O._LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    O.lib()
    return O


@pytest.fixture(scope="session")
def sift_lib():
    """The HIP library; built with hipcc (cross-compiles without a GPU) if absent."""
    import sift_amd
    if not os.path.exists(sift_amd.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sift_amd", "csrc"), "-j8", "-s"])
    sift_amd.lib()
    return sift_amd


def _load_rgb(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGB"))


@pytest.fixture(scope="session")
def lenna_bgr():
    return _load_rgb("Lenna.png")[:, :, ::-1].copy()


@pytest.fixture(scope="session")
def lenna_rotated_bgr():
    return _load_rgb("LennaRotated.png")[:, :, ::-1].copy()


@pytest.fixture(scope="session")
def school_pair(oracle):
    """The reference's two-image panorama example (res/school_left.jpg, res/school_right.jpg) through the
    CPU oracle: ((keypoints, descriptors) of left, of right).  Decoded with Pillow (libjpeg-turbo), gray by the
    oracle's 15-bit formula."""
    out = []
    for n in ("school_left.jpg", "school_right.jpg"):
        o = oracle.OracleSift(oracle.bgr_to_gray(_load_rgb(n)[:, :, ::-1].copy())).exec(threads=8, blur_threads=8)
        out.append((o.keypoints(), o.descriptors()))
    return out


@pytest.fixture(scope="session")
def known():
    import json
    return json.load(open(os.path.join(GOLDEN, "known_answers.json")))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + "_oracle.npz"))
    return load


@pytest.fixture(scope="session")
def ctx(sift_lib):
    c = sift_lib.Context(0)
    yield c
    c.close()
