"""Minimal driver for rocprofv3: two exec() batches of 1080p frames, nothing else."""
import sys
sys.path.insert(0, '.')
import numpy as np
import sift_amd as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
ctx = S.Context(0)
b = ctx.batch(W, H, B)
for i in range(B):
    b.upload(i, S.synth_dead_leaves(1, i, W, H))
for _ in range(2):
    b.run(B)
ctx.synchronize()
print(b.counts(B)['keypoints'].mean())
