"""Writes tests/golden/homography_cases.json from the CPU restatement (oracle/homography_oracle.c):
a regression fixture for the cases of tests/test_homography.py.  It does not pin parity with
OpenCV (which is not installed here); it only records what the restatement produced."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from test_homography import CASES, make_case  # noqa: E402

out = []
for case in CASES:
    src, dst = make_case(*case)
    H, mask, info = O.find_homography(src, dst)
    out.append({"case": list(case), "info": info, "inliers": int(mask.sum()),
                "H_hex": [float.hex(float(v)) for v in H.ravel()]})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "homography_cases.json"), "w"), indent=1)
print("wrote", len(out), "cases")
