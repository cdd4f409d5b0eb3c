"""One workspace, many passes: does the memory-bound blur level change speed over time within an allocation?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import sift_amd as S
B = 16
ctx = S.Context(0)
frames = np.stack([S.synth_dead_leaves(1, i, 1920, 1080) for i in range(B)])
pool = torch.from_numpy(frames).cuda()
for t in range(4):
    b = ctx.batch(1920, 1080, B)
    b.set_external_frames(pool.data_ptr())
    row = []
    t0 = time.time()
    for k in range(40):
        prof = b.run_profiled(B)
        row.append('%.2f/%.2f' % (prof['blur_l1']['ms'], prof['blur_l5']['ms']))
        if k == 19:
            time.sleep(1.0)
    print(t, '%.1fs' % (time.time() - t0), ' '.join(row), flush=True)
    b.close()
