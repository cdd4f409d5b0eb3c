"""sift_amd/host/image_io: PNG / JPEG / PNM files -> 8-bit BGR, the way cv::imread hands images to the
reference's main.cpp:30.  PNG: checked against PIL on the reference's own pictures and against
hand-built PNGs of every colour type, bit depth and scanline filter.  JPEG: checked bit for bit
against what libjpeg-turbo (through Pillow, scripts/make_jpeg_fixtures.py) decodes, on small files of
every coding variant and, where /root/reference is present, on all of the reference's res/*.jpg."""
import glob
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HOST = os.path.join(ROOT, "sift_amd", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("imgio") / "image_io_check")
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "image_io_check.cpp"), os.path.join(HOST, "image_io.cpp"),
                           os.path.join(HOST, "jpeg_decode.cpp"), "-lz", "-o", exe])

    def run(*paths):
        out = subprocess.run([exe, *map(str, paths)], check=True, capture_output=True, text=True).stdout
        return [line.split()[1:] for line in out.strip().splitlines()]
    run.exe = exe
    return run


def fnv(bgr):
    h = 1469598103934665603
    for b in bgr.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def expect(bgr):
    return [str(bgr.shape[1]), str(bgr.shape[0]), str(int(bgr.sum(dtype=np.uint64))), str(fnv(bgr))]


def chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def make_png(w, h, ctype, depth, lines, filters, palette=None, idat_split=1):
    """lines: h byte strings (unfiltered scanlines); filters: one filter type per line."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, channels * depth // 8)
    raw = bytearray()
    prev = bytes(len(lines[0]))
    for line, ft in zip(lines, filters):
        out = bytearray(len(line))
        for i, v in enumerate(line):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            out[i] = (v - pred) & 255
        raw.append(ft)
        raw += out
        prev = line
    z = zlib.compress(bytes(raw), 6)
    parts = [z[i * len(z) // idat_split:(i + 1) * len(z) // idat_split] for i in range(idat_split)]
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        png += chunk(b"PLTE", bytes(palette))
    png += chunk(b"tEXt", b"Comment\x00ancillary chunks are skipped")
    for p in parts:
        png += chunk(b"IDAT", p)
    return png + chunk(b"IEND", b"")


def test_reference_pictures_decode_like_pil(checker, lenna_bgr, lenna_rotated_bgr, known):
    got = checker(os.path.join(GOLDEN, "Lenna.png"), os.path.join(GOLDEN, "LennaRotated.png"))
    assert got[0] == expect(lenna_bgr)
    assert got[1] == expect(lenna_rotated_bgr)


@pytest.mark.parametrize("ctype,depth", [(0, 8), (0, 16), (0, 4), (0, 2), (0, 1), (2, 8), (2, 16), (3, 8), (3, 4), (3, 1),
                                         (4, 8), (4, 16), (6, 8), (6, 16)])
def test_every_colour_type_depth_and_filter(checker, tmp_path, ctype, depth):
    rng = np.random.default_rng(100 * ctype + depth)
    w, h = 37, 23  # odd sizes: partial bytes at the end of sub-byte scanlines
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    palette = rng.integers(0, 256, size=3 * (1 << min(depth, 8)), dtype=np.uint8) if ctype == 3 else None
    samples = rng.integers(0, 1 << depth, size=(h, w, channels), dtype=np.uint32)
    lines = []
    for y in range(h):
        if depth == 8:
            lines.append(samples[y].astype(np.uint8).tobytes())
        elif depth == 16:
            lines.append(samples[y].astype(">u2").tobytes())
        else:
            bits = np.zeros((w * depth + 7) // 8 * 8, np.uint8)
            vals = samples[y, :, 0]
            for k in range(depth):
                bits[np.arange(w) * depth + k] = (vals >> (depth - 1 - k)) & 1
            lines.append(np.packbits(bits).tobytes())
    filters = [(y + ctype) % 5 for y in range(h)]
    path = tmp_path / ("t_%d_%d.png" % (ctype, depth))
    path.write_bytes(make_png(w, h, ctype, depth, lines, filters, palette, idat_split=3))
    s8 = samples >> 8 if depth == 16 else samples
    if ctype in (2, 6):
        rgb = s8[:, :, :3]
    elif ctype == 3:
        rgb = palette.reshape(-1, 3)[samples[:, :, 0]]
    else:
        g = s8[:, :, 0] if depth >= 8 else samples[:, :, 0] * 255 // ((1 << depth) - 1)
        rgb = np.stack([g, g, g], axis=2)
    bgr = np.ascontiguousarray(rgb[:, :, ::-1]).astype(np.uint8)
    assert checker(path)[0] == expect(bgr)


def test_pnm_and_unreadable_files(checker, tmp_path):
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(9, 14, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, size=(9, 14), dtype=np.uint8)
    (tmp_path / "c.ppm").write_bytes(b"P6\n# a comment\n14 9\n255\n" + rgb.tobytes())
    (tmp_path / "g.pgm").write_bytes(b"P5 14 9 255\n" + gray.tobytes())
    (tmp_path / "bad.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"garbage" * 10)
    (tmp_path / "short.ppm").write_bytes(b"P6 14 9 255\n" + rgb.tobytes()[:50])
    got = checker(tmp_path / "c.ppm", tmp_path / "g.pgm", tmp_path / "bad.png", tmp_path / "short.ppm", tmp_path / "missing.png")
    assert got[0] == expect(np.ascontiguousarray(rgb[:, :, ::-1]))
    assert got[1] == expect(np.stack([gray] * 3, axis=2))
    assert got[2] == ["EMPTY"] and got[3] == ["EMPTY"] and got[4] == ["EMPTY"]


# ---- JPEG -------------------------------------------------------------------------------------

JPEG_DIR = os.path.join(GOLDEN, "jpeg")
JPEG_EXPECTED = json.load(open(os.path.join(JPEG_DIR, "expected.json")))


def as_list(e):
    return [str(e["w"]), str(e["h"]), str(e["sum"]), str(e["hash"])]


def test_jpeg_variants_decode_like_libjpeg_turbo(checker):
    """Baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, odd and tiny sizes (narrow components take the
    non-interpolating upsampler), optimised Huffman tables, 16-bit-range coefficients (quality 100) and
    coarse ones (quality 10), restart intervals, grey, and the eight EXIF orientations."""
    names = sorted(JPEG_EXPECTED["generated"])
    assert len(names) >= 32
    got = checker(*[os.path.join(JPEG_DIR, n) for n in names])
    for n, g in zip(names, got):
        assert g == as_list(JPEG_EXPECTED["generated"][n]), n


def test_jpeg_fixture_expectations_still_match_pillow():
    """The expectations are what Pillow's libjpeg-turbo decodes today, with cv::imread's EXIF handling."""
    from PIL import Image
    for n in ("base_420_odd.jpg", "prog_422_odd.jpg", "exif_6.jpg", "grey_prog.jpg"):
        im = Image.open(os.path.join(JPEG_DIR, n))
        o = im.getexif().get(0x0112) or 1
        bgr = np.array(im.convert("RGB"))[:, :, ::-1]
        if o == 6:  # transpose, then flip horizontally: a quarter turn clockwise
            bgr = bgr.transpose(1, 0, 2)[:, ::-1]
        assert expect(np.ascontiguousarray(bgr)) == as_list(JPEG_EXPECTED["generated"][n]), n


@pytest.mark.skipif(not os.path.isdir("/root/reference/res"), reason="the reference checkout is not on this machine")
def test_reference_jpegs_decode_like_libjpeg_turbo(checker):
    """All twenty res/*.jpg|jpeg of the reference (baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, a restart
    interval, an Adobe marker, up to 2000 x 3008), against hashes of Pillow's decode recorded by
    scripts/make_jpeg_fixtures.py."""
    names = sorted(JPEG_EXPECTED["reference_res"])
    assert len(names) == 20
    got = checker(*[os.path.join("/root/reference/res", n) for n in names])
    for n, g in zip(names, got):
        assert g == as_list(JPEG_EXPECTED["reference_res"][n]), n


def test_broken_and_unsupported_jpegs_yield_empty(checker, tmp_path):
    good = open(os.path.join(JPEG_DIR, "base_420.jpg"), "rb").read()
    sof = good.index(b"\xff\xc0")
    cases = {
        "not.jpg": b"\xff\xd8" + b"\x00" * 40,                      # SOI, then nothing usable
        "noscan.jpg": good[:sof] + b"\xff\xd9",                     # tables only
        "arith.jpg": good[:sof + 1] + b"\xc9" + good[sof + 2:],      # SOF9: arithmetic coding
        "p12.jpg": good[:sof + 4] + b"\x0c" + good[sof + 5:],        # 12-bit samples
        "cmyk.jpg": good[:sof + 9] + b"\x04" + good[sof + 10:],      # four components
        "cutseg.jpg": good[:sof + 6],                                # frame header cut short
    }
    for n, b in cases.items():
        (tmp_path / n).write_bytes(b)
    got = checker(*[tmp_path / n for n in cases])
    assert all(g == ["EMPTY"] for g in got), got
    # a file cut in the middle of its entropy-coded data still yields a full-sized picture (zeros are fed), as libjpeg does
    (tmp_path / "cut.jpg").write_bytes(good[:len(good) * 2 // 3])
    g = checker(tmp_path / "cut.jpg")[0]
    assert g[:2] == as_list(JPEG_EXPECTED["generated"]["base_420.jpg"])[:2]


def test_imwrite_png_round_trip(checker, tmp_path, lenna_bgr):
    """sift_io::imwrite (cv::imwrite at main.cpp:149): the PNG it writes decodes, with Pillow and with
    sift_io::imread, to the pixels it was given."""
    from PIL import Image
    out = tmp_path / "copy.png"
    subprocess.check_call([checker.exe, "--copy", os.path.join(GOLDEN, "Lenna.png"), str(out)])
    assert np.array_equal(np.array(Image.open(out).convert("RGB"))[:, :, ::-1], lenna_bgr)
    assert checker(out)[0] == expect(lenna_bgr)
    odd = tmp_path / "odd.png"
    subprocess.check_call([checker.exe, "--copy", os.path.join(JPEG_DIR, "tiny_3x5.jpg"), str(odd)])
    assert checker(odd)[0] == as_list(JPEG_EXPECTED["generated"]["tiny_3x5.jpg"])
    assert subprocess.call([checker.exe, "--copy", str(tmp_path / "missing.png"), str(tmp_path / "x.png")]) == 1


def test_mutated_files_do_not_break_the_decoders(tmp_path):
    """400 corrupted copies of the JPEG fixtures and of a PNG (random byte edits in headers and entropy data,
    truncations, absurd dimensions) through the decoders built with AddressSanitizer and UBSan: any result is
    acceptable, an out-of-bounds access or undefined operation is not."""
    import random
    exe = str(tmp_path / "image_io_check_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++20", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + HOST, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "image_io_check.cpp"),
                           os.path.join(HOST, "image_io.cpp"), os.path.join(HOST, "jpeg_decode.cpp"), "-lz", "-o", exe])
    rnd = random.Random(2024)
    sources = sorted(glob.glob(os.path.join(JPEG_DIR, "*.jpg")))
    png = open(os.path.join(GOLDEN, "Lenna.png"), "rb").read()[:20000]
    paths = []
    for k in range(400):
        if k % 8 == 7:
            b, ext = bytearray(png), ".png"
            for _ in range(rnd.randint(1, 6)):
                b[rnd.randrange(8, len(b) if rnd.random() < 0.5 else 60)] = rnd.randrange(256)
        else:
            b, ext = bytearray(open(rnd.choice(sources), "rb").read()), ".jpg"
            sos, mode = b.find(b"\xff\xda"), rnd.random()
            if mode < 0.4:
                for _ in range(rnd.randint(1, 4)):
                    b[rnd.randrange(2, sos + 14)] = rnd.randrange(256)
            elif mode < 0.7:
                for _ in range(rnd.randint(1, 30)):
                    b[rnd.randrange(sos, len(b))] = rnd.choice([0xFF, 0x00, rnd.randrange(256)])
            elif mode < 0.85:
                b = b[:rnd.randrange(2, len(b))]
            else:
                i = max(b.find(b"\xff\xc0"), b.find(b"\xff\xc2"))
                b[i + 5:i + 9] = struct.pack(">HH", rnd.choice([0, 1, 9, 65535, rnd.randrange(65536)]),
                                             rnd.choice([0, 1, 9, 65535, rnd.randrange(65536)]))
        p = tmp_path / ("m%03d%s" % (k, ext))
        p.write_bytes(bytes(b))
        paths.append(str(p))
    for i in range(0, len(paths), 100):
        r = subprocess.run([exe] + paths[i:i + 100], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]


def test_keypoint_and_match_overlays(tmp_path):
    """The stand-ins for cv::drawKeypoints(DRAW_RICH_KEYPOINTS) (sift.cpp:89) and cv::drawMatches (main.cpp:110),
    built with the address and undefined-behaviour sanitizers: circles and lines partly or wholly outside the
    canvas are clipped, the gray ground is kept, coloured pixels appear, and both pictures are valid PNG files."""
    from PIL import Image
    exe = str(tmp_path / "draw_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++20", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + HOST, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "draw_check.cpp"),
                           os.path.join(HOST, "image_io.cpp"), os.path.join(HOST, "jpeg_decode.cpp"), "-lz", "-o", exe])
    a, b = tmp_path / "kp.png", tmp_path / "m.png"
    out = subprocess.run([exe, str(a), str(b)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["160", "120", "320", "120"]
    kp, m = np.array(Image.open(a)), np.array(Image.open(b))
    assert kp.shape == (120, 160, 3) and m.shape == (120, 320, 3)
    yy, xx = np.mgrid[0:120, 0:160]
    gray = ((xx + yy) % 256).astype(np.uint8)
    coloured = (kp[:, :, 0] != kp[:, :, 1]) | (kp[:, :, 1] != kp[:, :, 2])
    assert 200 < coloured.sum() < 2000                       # three outlines and two radii, one pixel wide
    assert np.array_equal(kp[~coloured][:, 0], gray[~coloured])
    assert coloured[50, 40 + 15] or coloured[50, 40 - 15]    # the circle of radius size/2 = 15 round (40, 50)
    both = (m[:, :, 0] != m[:, :, 1]) | (m[:, :, 1] != m[:, :, 2])
    assert both.sum() > coloured.sum() * 2                   # two copies of the overlay plus the match lines
