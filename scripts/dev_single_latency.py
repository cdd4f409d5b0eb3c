"""Wall time of the single-image call (sift_extract_gray8) and of a 1-frame batch run for a few frame sizes."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import sift_amd as S
ctx = S.Context(0)
for (w, h) in ((512, 512), (800, 600), (1920, 1080)):
    img = S.synth_dead_leaves(3, 0, w, h)
    ctx.extract(img)
    t = time.perf_counter()
    for _ in range(20):
        kp, d, info = ctx.extract(img)
    t1 = (time.perf_counter() - t) / 20
    b = ctx.batch(w, h, 1)
    b.upload(0, img)
    b.run(1); b.counts(1)
    t = time.perf_counter()
    for _ in range(20):
        b.run(1)
        b.counts(1)
    t2 = (time.perf_counter() - t) / 20
    prof = b.run_profiled(1)
    dev = sum(v['ms'] for v in prof.values())
    nl = sum(v['launches'] for v in prof.values())
    print('%dx%d keypoints %d  extract() %.2f ms  batch.run+counts %.2f ms  device time %.2f ms in %d launches' % (w, h, len(kp), t1 * 1e3, t2 * 1e3, dev, nl), flush=True)
    b.close()
