"""Do two streams of half batches overlap better when one runs half a step behind the other?  Frames per second
over 40 steps of 64 frames: one stream, two streams in step, two streams staggered by an extra 16-frame run."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import sift_amd as S
W, H, B, STEPS = 1920, 1080, 64, 40
host = np.stack([S.synth_dead_leaves(1, i, W, H) for i in range(B)])
pool = torch.from_numpy(host).cuda()
torch.cuda.synchronize()

def measure(nstreams, stagger):
    ctxs = [S.Context(0) for _ in range(nstreams)]
    Bs = B // nstreams
    bts = [c.batch(W, H, Bs) for c in ctxs]
    for k, bt in enumerate(bts):
        bt.set_external_frames(pool.data_ptr() + k * Bs * H * W)
        bt.run(Bs)
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    if stagger:
        for k in range(1, nstreams):
            bts[k].run(Bs * k // nstreams)   # extra work that puts stream k a fraction of a step behind
    for _ in range(STEPS):
        for bt in bts:
            bt.run(Bs)
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    extra = sum(Bs * k // nstreams for k in range(1, nstreams)) if stagger else 0
    for bt in bts:
        bt.close()
    return (STEPS * B + extra) / dt

for ns, st in ((1, False), (2, False), (2, True), (4, False), (4, True), (1, False)):
    print('streams %d stagger %d: %.1f frames/s' % (ns, st, measure(ns, st)), flush=True)
