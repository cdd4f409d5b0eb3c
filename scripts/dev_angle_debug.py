import sys
sys.path.insert(0, '.')
import numpy as np
import sift_amd as S
from oracle import oracle as O
img = S.synth_dead_leaves(1, 0, 1920, 1080)
o = O.OracleSift(img).exec(threads=16, blur_threads=16)
ctx = S.Context(0)
kp, desc, info = ctx.extract(img)
okp = o.keypoints(); od = o.descriptors()
d = np.abs(kp['angle'].astype(np.float64) - okp['angle'])
bad = np.nonzero(d > 0)[0]
print('angle mismatches', len(bad), d[bad])
dd = np.abs(desc - od); print('desc rows mismatched', (dd.max(1) > 0).sum(), dd.max())
for i in bad[:5]:
    k = okp[i]; print(i, k, kp[i]['angle'])
    octv = (k['octave'] & 255); octv = octv - 256 if octv >= 128 else octv
    oct_ = octv + 1; layer = (k['octave'] >> 8) & 255
    L = o.gauss(oct_, layer); h, w = L.shape
    x = np.float32(k['x']) * 2; y = np.float32(k['y']) * 2; size = np.float32(k['size']) * 2
    bx = int(np.round(float(x) / (1 << oct_))); by = int(np.round(float(y) / (1 << oct_)))
    scale = 1.5 * float(size) / (1 << (oct_ + 1)); radius = int(np.round(scale * 3))
    near = []
    for i_ in range(-radius, radius + 1):
        for j_ in range(-radius, radius + 1):
            yy, xx = by + i_, bx + j_
            if 0 < yy < h - 1 and 0 < xx < w - 1:
                dx = L[yy, xx + 1] - L[yy, xx - 1]; dy = L[yy - 1, xx] - L[yy + 1, xx]
                ori = np.arctan2(dy, dx) * 180 / np.pi
                v = ori * 36 / 360
                fr = abs(v - np.floor(v) - 0.5)
                if fr < 1e-7 and abs(dx)+abs(dy) > 1e-6:
                    near.append((i_, j_, float(dx).hex(), float(dy).hex(), dx, dy, repr(float(ori)), fr))
    print('  boundary samples', len(near)); [print('   ', n) for n in near[:20]]
