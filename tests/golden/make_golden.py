"""Regenerates the golden vectors in this directory from the CPU oracle.

The reference itself cannot run here (it needs OpenCV, which is absent), so the
only reference-side pins are the published known answers kept in
known_answers.json (README.md:353-354 and the stage checkpoints recorded when
the reference's algorithm was first restated).  The .npz files are outputs of
oracle/sift_oracle.c on the committed input images; they let the GPU tests
compare against fixed data as well as against the live oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import oracle as O  # noqa: E402
import sift_amd as S  # noqa: E402  (host-only synthetic generator)


def run(gray):
    o = O.OracleSift(gray).exec(threads=8)
    c = o.candidates()
    return dict(keypoints=o.keypoints(), descriptors=o.descriptors().astype(np.uint8),
                candidates=np.stack([c["oct"], c["img"], c["i"], c["j"]], 1).astype(np.int32))


def main():
    for name in ("Lenna", "LennaRotated"):
        rgb = np.array(Image.open(os.path.join(HERE, name + ".png")).convert("RGB"))
        gray = O.bgr_to_gray(rgb[:, :, ::-1].copy())
        np.savez_compressed(os.path.join(HERE, name.lower() + "_oracle.npz"), gray=gray, **run(gray))
    leaves = S.synth_dead_leaves(7, 0, 320, 240)
    np.savez_compressed(os.path.join(HERE, "leaves_320x240_seed7_oracle.npz"), gray=leaves, **run(leaves))


if __name__ == "__main__":
    main()
