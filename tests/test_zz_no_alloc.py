"""sift_knn2_device_noalloc: the matcher entry point that allocates nothing (its scratch is a static device
array inside the library) must give bit-for-bit the results of sift_knn2_device, call after call, whatever the
previous call left in the shared scratch.  Collected last on purpose: a GPU run with -x stops at the first
failure, which would hide every test after this one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(sift_lib, ctx, nq, nt, base):
    import torch
    t = sift_lib.synth_descriptors(5, nt)
    q = sift_lib.synth_queries(6, t, nq)
    d_q, d_t = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    want = torch.full((nq, 4), -7, dtype=torch.int32, device="cuda")
    got = torch.full((nq, 4), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.knn2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, base, want.data_ptr())
    ctx.knn2_device_noalloc(d_q.data_ptr(), nq, d_t.data_ptr(), nt, base, got.data_ptr())
    ctx.synchronize()
    return want.cpu().numpy(), got.cpu().numpy(), q, t


def test_noalloc_matcher_equals_the_allocating_one_on_two_calls_in_a_row(sift_lib, ctx, oracle):
    # first call: several train splits (partial results go through the scratch), indices offset by a shard base
    want, got, q, t = _both(sift_lib, ctx, 3000, 50000, 1000)
    assert np.array_equal(got, want)
    i1, d1, i2, d2, _ = oracle.knn2_u8(q[:256], t, threads=16)
    assert np.array_equal(got[:256, 0], d1) and np.array_equal(got[:256, 1], i1 + 1000)
    assert np.array_equal(got[:256, 2], d2) and np.array_equal(got[:256, 3], i2 + 1000)
    # second call, smaller in both dimensions: the scratch still holds the first call's biases and partials
    want, got, q, t = _both(sift_lib, ctx, 700, 9000, 0)
    assert np.array_equal(got, want)
    i1, d1, i2, d2, _ = oracle.knn2_u8(q, t, threads=16)
    assert np.array_equal(got[:, 0], d1) and np.array_equal(got[:, 1], i1)
    assert np.array_equal(got[:, 2], d2) and np.array_equal(got[:, 3], i2)
    # a shape with a single train split (results written straight to the output), ragged tail
    want, got, _, _ = _both(sift_lib, ctx, 300001, 1000, 0)
    assert np.array_equal(got, want)


def test_noalloc_matcher_rejects_what_its_scratch_cannot_hold(sift_lib, ctx):
    import torch
    limit = 2097152
    # the train buffer really holds limit + 1 rows: were the call not rejected, it would still read valid memory
    train = torch.zeros((limit + 1, 128), dtype=torch.uint8, device="cuda")
    q = torch.zeros((512, 128), dtype=torch.uint8, device="cuda")
    out = torch.full((512, 4), -5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(sift_lib.SiftError) as e:   # rejected on the host before anything is launched
        ctx.knn2_device_noalloc(q.data_ptr(), 512, train.data_ptr(), limit + 1, 0, out.data_ptr())
    assert e.value.code == sift_lib.api.SIFT_ERR_INVALID and str(limit) in str(e.value)
    ctx.synchronize()
    assert (out.cpu().numpy() == -5).all()          # nothing was written
    ctx.knn2_device_noalloc(q.data_ptr(), 512, train.data_ptr(), limit, 0, out.data_ptr())  # the limit itself is served
    ctx.knn2_device_noalloc(q.data_ptr(), 0, train.data_ptr(), 512, 0, out.data_ptr())      # no queries: nothing to do
    ctx.synchronize()
    got = out.cpu().numpy()
    assert (got[:, 0] == 0).all() and (got[:, 1] == 0).all() and (got[:, 3] == 1).all()   # all-zero rows: ties go to rows 0, 1
