"""Wall time of the host-buffer entry points around extraction: matcher, homography, composition."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import sift_amd as S
ctx = S.Context(0)
t = S.synth_descriptors(5, 3000)
q = S.synth_queries(6, t, 3500)
def timeit(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3
print('knn2 3500 x 3000 (host buffers): %.3f ms' % timeit(lambda: ctx.knn2(q, t)))
t2 = S.synth_descriptors(5, 100000)
q2 = S.synth_queries(6, t2, 100000)
print('knn2 100000 x 100000 (host buffers): %.3f ms' % timeit(lambda: ctx.knn2(q2, t2), 5))
rng = np.random.default_rng(1)
H = np.array([[1.1, 0.05, 30], [-0.03, 0.95, -12], [1e-5, -2e-5, 1]])
src = (rng.random((300, 2)) * [1920, 1080]).astype(np.float32)
p = np.c_[src, np.ones(300)] @ H.T
dst = (p[:, :2] / p[:, 2:] + rng.normal(0, 0.3, (300, 2))).astype(np.float32)
dst[:90] = (rng.random((90, 2)) * [1920, 1080]).astype(np.float32)
print('find_homography 300 points: %.3f ms' % timeit(lambda: ctx.find_homography(src, dst)))
a = np.stack([S.synth_dead_leaves(1, 0, 800, 600)] * 3, axis=2)
b = np.stack([S.synth_dead_leaves(1, 1, 800, 600)] * 3, axis=2)
Hs = np.array([[0.74, 0.035, 501.0], [-0.12, 0.95, 4.7], [-3.2e-4, 2.8e-7, 1.0]])
print('stitch_pair 800x600 + 800x600: %.3f ms' % timeit(lambda: ctx.stitch_pair(a, b, Hs)))
