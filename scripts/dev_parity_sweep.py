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
for (w, h, seed, nf) in CASES:
    b = ctx.batch(w, h, 1)
    for f in range(nf):
        img = S.synth_dead_leaves(seed, f, w, h)
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
