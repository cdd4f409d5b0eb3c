"""Host arithmetic of the multi-GPU splits behind the C ABI (sift_shard_bounds, sift_knn2_exchange_plan): the
slices, paddings and offsets sift_knn2_sharded moves with RCCL, replayed here on the CPU with numpy copies as the
transport and the oracle as the per-GPU search and the merge.  No GPU call."""
import numpy as np
import pytest

IMAX = np.iinfo(np.int32).max


def test_c_shard_bounds_equal_the_python_ones(sift_lib):
    from sift_amd.dist import shard_bounds
    for n in (0, 1, 7, 4096, 10_000_000):
        for parts in (1, 2, 3, 8):
            assert sift_lib.shard_bounds(n, parts) == shard_bounds(n, parts)
    with pytest.raises(sift_lib.SiftError):
        sift_lib.shard_bounds(5, 0)


def test_exchange_plan_slices_partition_the_queries(sift_lib):
    for nq in (0, 1, 5, 101, 1 << 20):
        for parts in (1, 2, 3, 8):
            plans = [sift_lib.knn2_exchange_plan(nq, parts, r) for r in range(parts)]
            slice_len = plans[0][0]
            assert all(p[0] == slice_len for p in plans) and slice_len * parts >= nq > slice_len * parts - parts
            assert plans[0][1] == 0 and plans[-1][2] == nq
            assert all(a[2] == b[1] for a, b in zip(plans, plans[1:]))
            assert all(0 <= lo <= hi <= nq and hi - lo <= slice_len for _, lo, hi in plans)
    with pytest.raises(sift_lib.SiftError):
        sift_lib.knn2_exchange_plan(10, 4, 4)


def _merge(parts):
    nparts, n, _ = parts.shape
    out = np.empty((n, 4), np.int32)
    for k in range(n):
        c = [(int(parts[s, k, 0]), int(parts[s, k, 1])) for s in range(nparts) if parts[s, k, 1] >= 0]
        c += [(int(parts[s, k, 2]), int(parts[s, k, 3])) for s in range(nparts) if parts[s, k, 3] >= 0]
        c.sort()
        (d1, i1), (d2, i2) = (c + [(IMAX, -1), (IMAX, -1)])[:2]
        out[k] = (d1, i1, d2, i2)
    return out


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_sharded_matcher_layout_with_a_mock_transport(sift_lib, oracle, parts):
    nq, nt = 101, 900
    train = sift_lib.synth_descriptors(5, nt)
    queries = sift_lib.synth_queries(6, train, nq)
    i1, d1, i2, d2, _ = oracle.knn2_u8(queries, train, threads=4)
    bounds = sift_lib.shard_bounds(nt, parts)
    slice_len = sift_lib.knn2_exchange_plan(nq, parts, 0)[0]
    # per "GPU": partial results of all queries against its shard, global indices, padded with index -1 records
    part = np.full((parts, parts * slice_len, 4), -1, np.int32)
    for r in range(parts):
        a, b, c, d, _ = oracle.knn2_u8(queries, train[bounds[r]:bounds[r + 1]], threads=4)
        part[r, :nq] = np.stack([b, np.where(a >= 0, a + bounds[r], -1), d, np.where(c >= 0, c + bounds[r], -1)], 1)
    # the exchange: slice s of every part's buffer to part s, received in sender order
    recv = np.empty_like(part)
    for r in range(parts):
        for s in range(parts):
            recv[s, r * slice_len:(r + 1) * slice_len] = part[r, s * slice_len:(s + 1) * slice_len]
    got = np.full((nq, 4), -9, np.int32)
    for r in range(parts):
        _, lo, hi = sift_lib.knn2_exchange_plan(nq, parts, r)
        merged = _merge(recv[r].reshape(parts, slice_len, 4))
        got[lo:hi] = merged[:hi - lo]
    assert np.array_equal(got[:, 0], d1) and np.array_equal(got[:, 1], i1)
    assert np.array_equal(got[:, 2], d2) and np.array_equal(got[:, 3], i2)
