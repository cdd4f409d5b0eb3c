"""Minimal driver for rocprofv3: the matcher on device-resident descriptors."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import sift_amd as S
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = S.Context(0)
tr = S.synth_descriptors(5, nt); qq = S.synth_queries(6, tr, nq)
d_t = torch.from_numpy(tr).cuda(); d_q = torch.from_numpy(qq).cuda()
d_o = torch.empty((nq, 4), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
for _ in range(3):
    ctx.knn2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 0, d_o.data_ptr())
ctx.synchronize()
print(d_o[:2].cpu().numpy())
