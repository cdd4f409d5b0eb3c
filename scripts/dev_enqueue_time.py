import sys, time
sys.path.insert(0, '.')
import numpy as np
import sift_amd as S
ctx = S.Context(0)
for (w, h) in ((512, 512), (1920, 1080)):
    img = S.synth_dead_leaves(3, 0, w, h)
    b = ctx.batch(w, h, 1); b.upload(0, img); b.run(1); b.counts(1)
    te = ts = 0
    for _ in range(50):
        t0 = time.perf_counter(); b.run(1); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
        te += t1 - t0; ts += t2 - t1
    print(w, h, 'enqueue %.3f ms  wait %.3f ms' % (te / 50 * 1e3, ts / 50 * 1e3))
    b.close()
