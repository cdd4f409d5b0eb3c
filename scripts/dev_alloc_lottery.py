"""Allocates the 1080p batch workspace several times in one process and prints the per-stage device times of
each allocation: the blur levels bound by memory differ from one allocation to the next."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import sift_amd as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = S.Context(0)
frames = np.stack([S.synth_dead_leaves(1, i, 1920, 1080) for i in range(B)])
pool = torch.from_numpy(frames).cuda()
keep = []
for t in range(trials):
    b = ctx.batch(1920, 1080, B)
    b.set_external_frames(pool.data_ptr())
    for _ in range(3):
        prof = b.run_profiled(B)
    print(t, ' '.join('%s %.3f' % (k, prof[k]['ms']) for k in ('base', 'blur_l12', 'blur_l1', 'blur_l2', 'blur_l3', 'blur_l4', 'blur_l5', 'extrema', 'localize_orient', 'descriptor')), flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == 'keep':
        keep.append(b)
    else:
        b.close()
