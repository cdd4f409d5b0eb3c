"""Multi-process tests (gloo on the CPU, world sizes 2 and 3) of the N > 1 paths:
frame sharding and the sharded matcher's exchange."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_shard_bounds_cover_everything(sift_lib):
    from sift_amd.dist import frames_for_rank, shard_bounds
    for n in (0, 1, 7, 4096):
        for world in (1, 2, 3, 8):
            b = shard_bounds(n, world)
            assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
            got = [i for r in range(world) for i in frames_for_rank(n, r, world)]
            assert got == list(range(n))
            sizes = [len(frames_for_rank(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,port", [(2, 29621), (3, 29622)])
def test_sharded_matcher_is_independent_of_world_size(sift_lib, oracle, tmp_path, world, port):
    nq, nt = 101, 1000           # nq not divisible by the world size: padded last slice
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"),
           str(tmp_path), str(nq), str(nt)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    train = sift_lib.synth_descriptors(5, nt)
    queries = sift_lib.synth_queries(6, train, nq)
    i1, d1, i2, d2, acc = oracle.knn2_u8(queries, train, threads=4)
    slice_len = (nq + world - 1) // world
    frames = []
    for r in range(world):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        k = z["knn"]
        assert np.array_equal(k[:, 0], d1) and np.array_equal(k[:, 1], i1)
        assert np.array_equal(k[:, 2], d2) and np.array_equal(k[:, 3], i2)
        assert np.array_equal(z["acc"], acc)
        lo, hi = z["span_slice"]
        assert (lo, hi) == (r * slice_len, min(nq, (r + 1) * slice_len))
        assert np.array_equal(z["knn_slice"][:, 1], i1[lo:hi]) and np.array_equal(z["acc_slice"], acc[lo:hi])
        frames += z["frames"].tolist()
    assert frames == list(range(37))
