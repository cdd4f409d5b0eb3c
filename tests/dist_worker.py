"""Worker of tests/test_dist.py: one rank of a gloo process group on the CPU.

Exercises sift_amd.dist.sharded_knn2's exchange (slicing, padding, all-to-all,
gather) with CPU stand-ins for the two HIP kernels: the oracle's brute force for
the local top-2 and a numpy merge.  Writes this rank's result to an .npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import sift_amd  # noqa: E402
from sift_amd.dist import frames_for_rank, shard_bounds, sharded_knn2  # noqa: E402
from oracle import oracle as O  # noqa: E402

IMAX = np.iinfo(np.int32).max


def local_knn(q, t, base):
    i1, d1, i2, d2, _ = O.knn2_u8(q.numpy(), t.numpy().reshape(-1, 128), threads=2)
    out = np.stack([d1, np.where(i1 >= 0, i1 + base, -1), d2, np.where(i2 >= 0, i2 + base, -1)], 1).astype(np.int32)
    return torch.from_numpy(out)


def merge(parts):
    p = parts.numpy()
    nparts, n, _ = p.shape
    out = np.empty((n, 4), np.int32)
    acc = np.zeros(n, np.uint8)
    for k in range(n):
        cands = [(int(p[s, k, 0]), int(p[s, k, 1])) for s in range(nparts) if p[s, k, 1] >= 0]
        cands += [(int(p[s, k, 2]), int(p[s, k, 3])) for s in range(nparts) if p[s, k, 3] >= 0]
        cands.sort()
        (d1, i1), (d2, i2) = (cands + [(IMAX, -1), (IMAX, -1)])[:2]
        out[k] = (d1, i1, d2, i2)
        acc[k] = i2 >= 0 and np.sqrt(np.float32(d1)) < np.float32(0.7) * np.sqrt(np.float32(d2))
    return torch.from_numpy(out), torch.from_numpy(acc)


def hip_kernels():
    """The two HIP kernels of GPU 0 behind CPU tensors (every rank of the test shares the one GPU; the exchange itself
    stays on gloo)."""
    ctx = sift_amd.Context(0)

    def search(q, t, base):
        dq, dt = q.cuda(), t.cuda()
        out = torch.empty((q.shape[0], 4), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.knn2_device(dq.data_ptr(), q.shape[0], dt.data_ptr() if t.shape[0] else 0, t.shape[0], base, out.data_ptr())
        ctx.synchronize()
        return out.cpu()

    def merge_parts(parts):
        dp = parts.contiguous().cuda()
        nparts, n, _ = parts.shape
        out = torch.empty((n, 4), dtype=torch.int32, device="cuda")
        acc = torch.empty((n,), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.knn2_merge_device(dp.data_ptr(), nparts, n, out.data_ptr(), acc.data_ptr())
        ctx.synchronize()
        return out.cpu(), acc.cpu()
    return search, merge_parts


def main():
    global local_knn, merge
    outdir, nq, nt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "hip":
        local_knn, merge = hip_kernels()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    train = sift_amd.synth_descriptors(5, nt)
    queries = sift_amd.synth_queries(6, train, nq)
    b = shard_bounds(nt, world)
    shard = torch.from_numpy(train[b[rank]:b[rank + 1]].copy())
    knn, acc, span = sharded_knn2(torch.from_numpy(queries), shard, b[rank], local_knn=local_knn, merge=merge)
    knn_s, acc_s, span_s = sharded_knn2(torch.from_numpy(queries), shard, b[rank], local_knn=local_knn, merge=merge,
                                        gather=False)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), knn=knn.numpy(), acc=acc.numpy(), span=np.array(span),
             knn_slice=knn_s.numpy(), acc_slice=acc_s.numpy(), span_slice=np.array(span_s),
             frames=np.array(list(frames_for_rank(37, rank, world))))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
