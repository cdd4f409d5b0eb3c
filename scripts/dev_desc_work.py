"""Descriptor stage of a 16-frame 1080p batch: device time, contributing samples and LDS updates issued."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import sift_amd as S
B = 16
ctx = S.Context(0)
frames = np.stack([S.synth_dead_leaves(1, i, 1920, 1080) for i in range(B)])
pool = torch.from_numpy(frames).cuda()
b = ctx.batch(1920, 1080, B)
b.set_external_frames(pool.data_ptr())
for _ in range(3):
    prof = b.run_profiled(B)
w = b.descriptor_work(B)
print('descriptor %.3f ms  samples %.4g  updates %.4g  updates/sample %.2f' % (prof['descriptor']['ms'], w[:, 0].sum(), w[:, 1].sum(), w[:, 1].sum() / w[:, 0].sum()))
