"""Times the matcher on device-resident descriptors: median of several launches."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import sift_amd as S
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = S.Context(0)
tr = S.synth_descriptors(5, nt); qq = S.synth_queries(6, tr, nq)
d_t = torch.from_numpy(tr).cuda(); d_q = torch.from_numpy(qq).cuda()
d_o = torch.empty((nq, 4), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
ts = []
for _ in range(8):
    t0 = time.perf_counter()
    ctx.knn2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 0, d_o.data_ptr())
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
med = ts[len(ts) // 2]
print('knn %d x %d: median %.2f ms  min %.2f ms  %.1f%% of 5 PFLOP/s' % (nq, nt, med * 1e3, ts[0] * 1e3, 2 * 128 * nq * nt / med / 5e15 * 100))
