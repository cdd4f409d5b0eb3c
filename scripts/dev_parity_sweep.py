"""Sweep of synthetic frames: exact-equality statistics GPU vs oracle, single-frame schedule and batch schedule."""
import sys
sys.path.insert(0, '.')
import numpy as np
import sift_amd as S
from oracle import oracle as O
ctx = S.Context(0)
tot = dict(kp=0, angle=0, desc_rows=0, frames=0, count_mismatch=0, batch_schedule_differs=0)
import os
BIG = os.environ.get('SWEEP') == 'big'
CASES = [(1920, 1080, 1, 6), (960, 540, 1, 6), (640, 480, 9, 8)]
if BIG:
    CASES = [(1920, 1080, 3, 24), (1280, 720, 4, 24), (640, 480, 5, 40), (1001, 703, 6, 12), (3840, 2160, 7, 2), (333, 517, 8, 20)]
if os.environ.get('SWEEP') == 'big2':  # other seeds and sizes than 'big'
    CASES = [(1920, 1080, 11, 24), (1280, 720, 12, 24), (800, 600, 13, 30), (1023, 767, 14, 12), (3840, 2160, 15, 2),
             (517, 333, 16, 20), (2048, 1080, 2, 4), (1366, 768, 17, 12)]
def texture(kind, seed, f, w, h):
    """Frames that are not piecewise constant: white noise, low-pass noise, sums of oriented sinusoids."""
    rng = np.random.default_rng(1000 * seed + f)
    if kind == 'noise':
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == 'lowpass':
        k = int(rng.integers(3, 9))
        coarse = rng.integers(0, 256, (h // k + 2, w // k + 2)).astype(np.float64)
        up = np.kron(coarse, np.ones((k, k)))[:h + k, :w + k]
        c = np.cumsum(np.cumsum(np.pad(up, ((1, 0), (1, 0))), 0), 1)          # k x k box filter by summed-area table
        box = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
        return np.clip(np.rint(box[:h, :w] + rng.normal(0, 2, (h, w))), 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    acc = np.zeros((h, w))
    for _ in range(12):
        th, fr, ph, am = rng.uniform(0, np.pi), rng.uniform(0.01, 0.25), rng.uniform(0, 6.28), rng.uniform(5, 30)
        acc += am * np.sin(fr * (np.cos(th) * xx + np.sin(th) * yy) + ph)
    return np.clip(np.rint(128 + acc), 0, 255).astype(np.uint8)


TEXTURES = None
if os.environ.get('SWEEP') == 'textures':
    TEXTURES = True
    CASES = [(640, 480, 'noise', 6), (1280, 720, 'noise', 2), (640, 480, 'lowpass', 16), (1280, 720, 'lowpass', 8),
             (1920, 1080, 'lowpass', 4), (640, 480, 'waves', 12), (1001, 703, 'waves', 6), (1920, 1080, 'waves', 3)]
for (w, h, seed, nf) in CASES:
    b = ctx.batch(w, h, 1)
    for f in range(nf):
        img = texture(seed, 21, f, w, h) if TEXTURES else S.synth_dead_leaves(seed, f, w, h)
        o = O.OracleSift(img).exec(threads=16, blur_threads=16)
        kp, desc, info = ctx.extract(img)
        b.upload(0, img)
        b.run_profiled(1)                      # the batch schedule (fused blur levels, two-wave per-keypoint kernels)
        bkp, bdesc = b.download(0)
        if not (np.array_equal(bkp, kp) and np.array_equal(bdesc, desc.astype(np.uint8))):
            tot['batch_schedule_differs'] += 1
            print('BATCH SCHEDULE DIFFERS', w, h, seed, f)
        okp, od = o.keypoints(), o.descriptors()
        tot['frames'] += 1
        if len(kp) != len(okp):
            tot['count_mismatch'] += 1
            print('COUNT MISMATCH', w, h, seed, f, len(kp), len(okp))
            continue
        for fld in ('x', 'y', 'size', 'response', 'octave'):
            assert np.array_equal(kp[fld], okp[fld]), fld
        na = int((kp['angle'] != okp['angle']).sum())
        nd = int((np.abs(desc - od).max(1) > 0).sum())
        tot['kp'] += len(kp); tot['angle'] += na; tot['desc_rows'] += nd
        if na or nd:
            print(w, h, seed, f, 'angle mismatches', na, 'max', np.abs(kp['angle'].astype(np.float64) - okp['angle']).max(), 'desc rows', nd)
    b.close()
print(tot)
