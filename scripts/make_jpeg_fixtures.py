"""Writes tests/golden/jpeg/*.jpg and tests/golden/jpeg/expected.json.

The JPEG files are encoded here with Pillow from tests/golden/Lenna.png (crops and resizes) to cover
the coding variants; the expected pixels are what Pillow's bundled libjpeg-turbo decodes them to at its
default settings (islow DCT, fancy upsampling), which is the decoder cv::imread uses, plus the EXIF
orientation applied as cv::imread applies it.  When /root/reference is present, the same is recorded
for the reference's own res/*.jpg files (hashes only; the files themselves are not copied)."""
import glob
import io
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
OUT = os.path.join(ROOT, "tests", "golden", "jpeg")
os.makedirs(OUT, exist_ok=True)


def fnv(a):
    h = 1469598103934665603
    for b in a.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def orient(a, o):
    """cv::imread: 2 flip x; 3 flip both; 4 flip y; 5 transpose; 6 transpose, flip x; 7 transpose, flip both;
    8 transpose, flip y."""
    if o >= 5:
        a = a.transpose(1, 0, 2)
    if o in (2, 3, 6, 7):
        a = a[:, ::-1]
    if o in (3, 4, 7, 8):
        a = a[::-1]
    return np.ascontiguousarray(a)


def expected(path):
    im = Image.open(path)
    o = im.getexif().get(0x0112) or 1
    bgr = np.array(im.convert("RGB"))[:, :, ::-1]
    bgr = orient(bgr, o if 1 <= o <= 8 else 1)
    return {"w": int(bgr.shape[1]), "h": int(bgr.shape[0]), "sum": int(bgr.astype(np.uint64).sum()), "hash": fnv(bgr)}


def exif_with_orientation(o):
    ex = Image.Exif()
    ex[0x0112] = o
    return ex.tobytes()


lenna = Image.open(os.path.join(ROOT, "tests", "golden", "Lenna.png")).convert("RGB")
noise = Image.fromarray(np.random.default_rng(7).integers(0, 256, (40, 56, 3), dtype=np.uint8).astype(np.uint8))
cases = {
    "base_444": (lenna.crop((100, 100, 228, 196)), dict(quality=90, subsampling=0)),
    "base_422": (lenna.crop((100, 100, 228, 196)), dict(quality=85, subsampling=1)),
    "base_420": (lenna.crop((100, 100, 228, 196)), dict(quality=75, subsampling=2)),
    "base_420_odd": (lenna.crop((60, 40, 161, 117)), dict(quality=80, subsampling=2)),
    "base_422_odd": (lenna.crop((60, 40, 161, 117)), dict(quality=80, subsampling=1)),
    "base_420_optimized": (lenna.crop((200, 200, 333, 290)), dict(quality=60, subsampling=2, optimize=True)),
    "base_420_q10": (lenna.crop((200, 200, 296, 264)), dict(quality=10, subsampling=2)),
    "base_444_q100": (lenna.crop((200, 200, 264, 248)), dict(quality=100, subsampling=0)),
    "base_420_restart": (lenna.crop((0, 0, 150, 100)), dict(quality=80, subsampling=2, restart_marker_blocks=3)),
    "base_444_restart_rows": (lenna.crop((0, 0, 90, 70)), dict(quality=80, subsampling=0, restart_marker_rows=1)),
    "prog_420": (lenna.crop((256, 256, 384, 352)), dict(quality=85, subsampling=2, progressive=True)),
    "prog_444": (lenna.crop((256, 256, 384, 352)), dict(quality=92, subsampling=0, progressive=True)),
    "prog_422_odd": (lenna.crop((256, 256, 355, 333)), dict(quality=70, subsampling=1, progressive=True)),
    "prog_420_restart": (lenna.crop((10, 300, 140, 400)), dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=2)),
    "prog_noise": (noise, dict(quality=95, subsampling=2, progressive=True)),
    "base_noise": (noise, dict(quality=95, subsampling=2)),
    "grey_base": (lenna.crop((100, 100, 201, 177)).convert("L"), dict(quality=85)),
    "grey_prog": (lenna.crop((100, 100, 201, 177)).convert("L"), dict(quality=85, progressive=True)),
    "tiny_1x1": (lenna.crop((5, 5, 6, 6)), dict(quality=90, subsampling=2)),
    "tiny_2x2": (lenna.crop((5, 5, 7, 7)), dict(quality=90, subsampling=2)),
    "tiny_3x5": (lenna.crop((5, 5, 8, 10)), dict(quality=90, subsampling=2)),
    "tiny_4x3_422": (lenna.crop((5, 5, 9, 8)), dict(quality=90, subsampling=1)),
    "tiny_5x17": (lenna.crop((5, 5, 10, 22)), dict(quality=90, subsampling=2)),
    "wide_17x16": (lenna.crop((5, 5, 22, 21)), dict(quality=90, subsampling=2)),
}
for o in range(1, 9):
    cases["exif_%d" % o] = (lenna.crop((300, 100, 348, 132)), dict(quality=90, subsampling=2, exif=exif_with_orientation(o)))

exp = {}
for name, (img, kw) in cases.items():
    path = os.path.join(OUT, name + ".jpg")
    img.save(path, "JPEG", **kw)
    exp[name + ".jpg"] = expected(path)

ref = {}
for f in sorted(glob.glob("/root/reference/res/**/*.jp*g", recursive=True)):
    ref[f.split("/res/")[1]] = expected(f)
json.dump({"generated": exp, "reference_res": ref}, open(os.path.join(OUT, "expected.json"), "w"), indent=1)
print(len(exp), "generated,", len(ref), "reference files;", sum(os.path.getsize(os.path.join(OUT, n)) for n in exp), "bytes")
