"""Minimal driver for rocprofv3: a few single-image runs of one frame size, nothing else."""
import sys
sys.path.insert(0, '.')
import sift_amd as S
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
ctx = S.Context(0)
b = ctx.batch(w, h, 1)
b.upload(0, S.synth_dead_leaves(3, 0, w, h))
for _ in range(4):
    b.run(1)
    ctx.synchronize()
