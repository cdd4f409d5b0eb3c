"""GPU tests of the single-process multi-GPU entry points (sift_group_*): on a one-GPU box the group has one member,
which still takes the whole path -- RCCL communicator, grouped send/receive of the query slices (to itself), merge,
the double-buffered host-to-host extraction pipeline -- and must reproduce the single-context calls bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group(sift_lib, ctx):
    g = sift_lib.Group([ctx])
    yield g
    g.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (101, 1000), (5000, 33000), (7, 0)])
def test_sharded_matcher_equals_the_single_context_one(sift_lib, ctx, group, oracle, nq, nt):
    t = sift_lib.synth_descriptors(5, nt)
    q = sift_lib.synth_queries(6, t, nq) if nt else np.zeros((nq, 128), np.uint8)
    want, wacc = ctx.knn2(q, t)
    got, gacc = group.knn2(q, t)
    assert np.array_equal(got, want) and np.array_equal(gacc, wacc)
    if nt:
        i1, d1, i2, d2, oacc = oracle.knn2_u8(q, t, threads=16)
        assert np.array_equal(got["i1"], i1) and np.array_equal(got["d1"], d1)
        assert np.array_equal(got["i2"], i2) and np.array_equal(got["d2"], d2) and np.array_equal(gacc, oacc)


@pytest.mark.parametrize("members", [2, 3, 8])
def test_sharded_matcher_with_several_members_on_one_gpu(sift_lib, ctx, oracle, members):
    """More than one member: train shards, per-member feed threads, slice offsets on the send side, sender-order
    receive, merge of `members` parts on every slice owner, per-owner download.  The members share the only GPU of
    this box, so the slices travel as device-to-device copies; query counts that are not a multiple of the slice,
    fewer train rows than members, fewer queries than members."""
    others = [sift_lib.Context(0) for _ in range(members - 1)]
    g = sift_lib.Group([ctx] + others)
    try:
        for nq, nt in [(1001, 20000), (members - 1, 977), (5, max(1, members - 2)), (64 * members + 1, 4096), (3, 0)]:
            t = sift_lib.synth_descriptors(5, nt)
            q = sift_lib.synth_queries(6, t, nq) if nt else np.zeros((nq, 128), np.uint8)
            want, wacc = ctx.knn2(q, t)
            for _ in range(2):  # the second call reuses the group's buffers and events
                got, gacc = g.knn2(q, t)
                assert np.array_equal(got, want) and np.array_equal(gacc, wacc)
            if nt:
                i1, d1, i2, d2, oacc = oracle.knn2_u8(q, t, threads=16)
                assert np.array_equal(got["i1"], i1) and np.array_equal(got["d1"], d1)
                assert np.array_equal(got["i2"], i2) and np.array_equal(got["d2"], d2) and np.array_equal(gacc, oacc)
    finally:
        g.close()
        for o in others:
            o.close()


def test_groups_over_all_gpus_of_the_box(sift_lib, ctx, oracle):
    """One context per GPU (RCCL transport for the matcher, one pipeline thread per GPU for the frames)."""
    ngpu = min(sift_lib.device_count(), 4)
    if ngpu < 2:
        pytest.skip("needs at least two GPUs")
    ctxs = [sift_lib.Context(i) for i in range(ngpu)]
    g = sift_lib.Group(ctxs)
    try:
        for nq, nt in [(1001, 20000), (ngpu - 1, 977), (5, ngpu - 1)]:
            t = sift_lib.synth_descriptors(5, nt)
            q = sift_lib.synth_queries(6, t, nq)
            want, wacc = ctx.knn2(q, t)
            got, gacc = g.knn2(q, t)
            assert np.array_equal(got, want) and np.array_equal(gacc, wacc)
        w, h, n = 320, 240, 2 * ngpu + 1
        frames = np.stack([sift_lib.synth_dead_leaves(7, i, w, h) for i in range(n)])
        want = [ctx.extract(f) for f in frames]
        for (kp, desc, _), (wkp, wdesc, _) in zip(g.extract_batch(frames, 2), want):
            assert np.array_equal(kp, wkp) and np.array_equal(desc, wdesc.astype(np.uint8))
    finally:
        g.close()
        for c in ctxs:
            c.close()


def test_sharded_matcher_rejects_bytes_above_127(sift_lib, group):
    with pytest.raises(sift_lib.SiftError) as e:
        group.knn2(np.full((4, 128), 128, np.uint8), np.zeros((4, 128), np.uint8))
    assert e.value.code == sift_lib.api.SIFT_ERR_NOT_INTEGRAL


def test_batch_extraction_from_host_frames_equals_single_calls(sift_lib, ctx, group):
    w, h, n = 320, 240, 7
    frames = np.stack([sift_lib.synth_dead_leaves(7, i, w, h) for i in range(n)])
    want = [ctx.extract(f) for f in frames]
    for per_launch in (3, 16):          # chunks of 3, 3, 1 through both workspaces; then one chunk
        got = group.extract_batch(frames, per_launch)
        assert len(got) == n
        for (kp, desc, info), (wkp, wdesc, winfo) in zip(got, want):
            assert np.array_equal(kp, wkp) and np.array_equal(desc, wdesc.astype(np.uint8))
            assert info["candidates"] == winfo["candidates"] and info["keypoints"] == winfo["keypoints"]
    # rows and frames with padding between them
    padded = np.zeros((n, h + 3, w + 5), np.uint8)
    padded[:, :h, :w] = frames
    import ctypes as C
    res = (sift_lib.api._FrameResult * n)()
    rc = sift_lib.lib().sift_extract_batch_u8(group._h, padded.ctypes.data, n, w, h, padded.strides[1], padded.strides[0], 4, res)
    assert rc == 0
    for r, (wkp, wdesc, _) in zip(res, want):
        assert r.n == len(wkp)
        d = np.ctypeslib.as_array(C.cast(r.descriptors, C.POINTER(C.c_uint8)), (r.n, 128))
        assert np.array_equal(d, wdesc.astype(np.uint8))
    sift_lib.lib().sift_frame_results_free(res, n)


def test_frames_split_over_two_members_driven_by_two_host_threads(sift_lib, ctx):
    """Two contexts in one group (both on the only GPU of this box): the frames are cut into two contiguous blocks,
    each driven by its own host thread and pipeline, and every result lands at its frame's index."""
    w, h, n = 320, 240, 9
    frames = np.stack([sift_lib.synth_dead_leaves(7, i, w, h) for i in range(n)])
    want = [ctx.extract(f) for f in frames]
    other = sift_lib.Context(0)
    g = sift_lib.Group([ctx, other])
    try:
        assert sift_lib.shard_bounds(n, 2) == [0, 4, 9]
        for per_launch in (2, 16):
            got = g.extract_batch(frames, per_launch)
            for (kp, desc, _), (wkp, wdesc, _) in zip(got, want):
                assert np.array_equal(kp, wkp) and np.array_equal(desc, wdesc.astype(np.uint8))
        got = g.extract_batch(frames[:1], 4)               # fewer frames than members: one of them gets none
        assert np.array_equal(got[0][0], want[0][0])
        with pytest.raises(sift_lib.SiftError):
            sift_lib.Group([ctx, ctx])
    finally:
        g.close()
        other.close()


def test_library_chosen_launch_sizes_cover_every_frame_once(sift_lib, ctx):
    """frames_per_launch = 0: the library cuts the run into a quarter-size launch, full ones and a quarter-size one
    (4, 16, 15, 4 frames here); every frame's result must land at its index, whatever launch carried it."""
    w, h, n = 160, 120, 39
    frames = np.stack([sift_lib.synth_dead_leaves(11, i, w, h) for i in range(n)])
    want = [ctx.extract(f) for f in frames]
    g = sift_lib.Group([ctx])
    try:
        for count in (n, 17, 16, 5):
            got = g.extract_batch(frames[:count], 0)
            assert len(got) == count
            for (kp, desc, _), (wkp, wdesc, _) in zip(got, want):
                assert np.array_equal(kp, wkp) and np.array_equal(desc, wdesc.astype(np.uint8))
    finally:
        g.close()


def test_a_smaller_launch_size_is_honoured_on_a_reused_group(sift_lib, ctx):
    """The group keeps the larger workspace of an earlier call, but launches must follow the size the caller names
    now: with frames_per_launch = 2 after a call with 8, seven frames take four launches, not one.  A negative size
    is a cap on launches whose sizes the library chooses."""
    w, h, n = 160, 120, 7
    frames = np.stack([sift_lib.synth_dead_leaves(11, i, w, h) for i in range(n)])
    want = [ctx.extract(f) for f in frames]
    g = sift_lib.Group([ctx])
    try:
        for per_launch, launches in ((8, 1), (2, 4), (0, 1), (-4, 4)):   # -4: at most 4, cut by the library: 1, 3, 2, 1
            before = ctx.pipeline_runs()
            got = g.extract_batch(frames, per_launch)
            assert ctx.pipeline_runs() - before == launches
            for (kp, desc, _), (wkp, wdesc, _) in zip(got, want):
                assert np.array_equal(kp, wkp) and np.array_equal(desc, wdesc.astype(np.uint8))
    finally:
        g.close()


def test_batch_extraction_reruns_frames_that_overflow_their_lists(sift_lib, ctx, oracle, lenna_bgr):
    gray = oracle.bgr_to_gray(lenna_bgr)
    frames = np.stack([gray, gray[::-1].copy()])
    want = [ctx.extract(f) for f in frames]
    g = sift_lib.Group([ctx])
    try:
        ctx.set_single_image_capacity(64)          # also the list capacity of the group's workspaces
        got = g.extract_batch(frames, 2)
        for (kp, desc, _), (wkp, wdesc, _) in zip(got, want):
            assert len(kp) > 64 and np.array_equal(kp, wkp) and np.array_equal(desc, wdesc.astype(np.uint8))
    finally:
        ctx.set_single_image_capacity(0)
        g.close()


def test_sharded_knn2_over_a_one_rank_rccl_group(sift_lib, ctx, oracle):
    """sift_amd.dist.sharded_knn2 with its default HIP search and merge inside a real `nccl` process group (one rank
    here): all_to_all_single and all_gather over RCCL, query count not a multiple of the slice."""
    import os
    import torch
    import torch.distributed as dist
    from sift_amd.dist import sharded_knn2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nq, nt = 1001, 20000
        t = sift_lib.synth_descriptors(5, nt)
        q = sift_lib.synth_queries(6, t, nq)
        i1, d1, i2, d2, oacc = oracle.knn2_u8(q, t, threads=16)
        # operands produced by asynchronous torch work right before the call
        d_t = torch.from_numpy(t).cuda().to(torch.int32).to(torch.uint8)
        d_q = torch.from_numpy(q).cuda().to(torch.int32).to(torch.uint8)
        for gather in (True, False):
            knn, acc, (lo, hi) = sharded_knn2(d_q, d_t, 0, ctx=ctx, gather=gather)
            k = knn.cpu().numpy()
            assert (lo, hi) == (0, nq)
            assert np.array_equal(k[:, 0], d1) and np.array_equal(k[:, 1], i1)
            assert np.array_equal(k[:, 2], d2) and np.array_equal(k[:, 3], i2)
            assert np.array_equal(acc.cpu().numpy(), oacc)
    finally:
        if started:
            dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29651), (3, 29652)])
def test_sharded_matcher_over_several_processes_with_the_hip_kernels(sift_lib, oracle, tmp_path, world, port):
    """Two and three ranks (separate processes that share this box's GPU): each rank's HIP search over its train shard
    with global indices, the exchange of query slices (gloo here: RCCL takes one rank per GPU), the HIP merge on each
    slice owner.  Equal to the unsharded CPU brute force whatever the number of ranks; the query count does not divide
    by it and the last train shard is ragged."""
    import os
    import subprocess
    import sys
    nq, nt = 1001, 30011
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "dist_worker.py"),
           str(tmp_path), str(nq), str(nt), "hip"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    train = sift_lib.synth_descriptors(5, nt)
    queries = sift_lib.synth_queries(6, train, nq)
    i1, d1, i2, d2, acc = oracle.knn2_u8(queries, train, threads=16)
    slice_len = (nq + world - 1) // world
    for r in range(world):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        k = z["knn"]
        assert np.array_equal(k[:, 0], d1) and np.array_equal(k[:, 1], i1)
        assert np.array_equal(k[:, 2], d2) and np.array_equal(k[:, 3], i2)
        assert np.array_equal(z["acc"], acc)
        lo, hi = z["span_slice"]
        assert (lo, hi) == (r * slice_len, min(nq, (r + 1) * slice_len))
        assert np.array_equal(z["knn_slice"][:, 1], i1[lo:hi]) and np.array_equal(z["acc_slice"], acc[lo:hi])
