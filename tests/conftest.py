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
