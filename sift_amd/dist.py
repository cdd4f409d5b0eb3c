"""Multi-GPU partitioning: one process per GPU over torch.distributed (RCCL).

Two ways the work splits across the GPUs of a node:

* Extraction: frames are independent, so a batch is cut into contiguous blocks
  of frames, one block per rank, with no data-path collective
  (``frames_for_rank``).
* Matching: every rank holds all queries and one contiguous shard of the train
  descriptors.  Each rank finds its local top-2 per query (indices offset to
  global train indices), then ONE all-to-all moves, for every query slice, the
  ``world`` partial results to the rank that owns the slice; that rank merges
  them under (distance, index) order and applies the ratio test.  The result
  does not depend on the number of ranks (``sharded_knn2``).

torch is used for process groups, collectives and device memory only; the top-2
search and the merge are HIP kernels reached through the C ABI.
"""
import numpy as np


def shard_bounds(n, world):
    """Contiguous near-equal split of range(n) into `world` parts: part r is [b[r], b[r+1])."""
    return [n * r // world for r in range(world + 1)]


def frames_for_rank(n_frames, rank, world):
    b = shard_bounds(n_frames, world)
    return range(b[rank], b[rank + 1])


def _hip_local_knn(ctx):
    import torch

    def run(q, t, train_base):
        out = torch.empty((q.shape[0], 4), dtype=torch.int32, device=q.device)
        ctx.knn2_device(q.data_ptr(), q.shape[0], t.data_ptr() if t.shape[0] else 0, t.shape[0], train_base,
                        out.data_ptr())
        return out
    return run


def _hip_merge(ctx):
    import torch

    def run(parts):
        nparts, n, _ = parts.shape
        out = torch.empty((n, 4), dtype=torch.int32, device=parts.device)
        acc = torch.empty((n,), dtype=torch.uint8, device=parts.device)
        ctx.knn2_merge_device(parts.data_ptr(), nparts, n, out.data_ptr(), acc.data_ptr())
        return out, acc
    return run


def _synchronised(ctx, fn):
    """The host-synchronised form, for a HIP kernel used beside an injected stand-in: torch's stream is drained
    before the engine's stream starts and the engine's before torch goes on."""
    import torch

    def run(*args):
        torch.cuda.current_stream().synchronize()
        out = fn(*args)
        ctx.synchronize()
        return out
    return run


class _on_engine_stream:
    """Runs a block of torch work on the context's own HIP stream: the engine's kernels, the tensors torch fills
    around them and the collectives are then ordered by that one stream, with no host synchronisation.  On entry the
    engine stream waits for the caller's stream (the inputs may still be in flight there); on exit the caller's stream
    waits for the engine stream, and the tensors handed back are marked as in use on the caller's stream."""

    def __init__(self, ctx):
        import torch
        self.torch = torch
        self.engine = torch.cuda.ExternalStream(ctx.stream, device=torch.device("cuda", ctx.device))
        self.caller = None
        self.guard = None

    def __enter__(self):
        self.caller = self.torch.cuda.current_stream(self.engine.device)
        self.engine.wait_stream(self.caller)
        self.guard = self.torch.cuda.stream(self.engine)
        self.guard.__enter__()
        return self

    def hand_back(self, *tensors):
        for t in tensors:
            t.record_stream(self.caller)
        return tensors

    def __exit__(self, *exc):
        self.guard.__exit__(*exc)
        self.caller.wait_stream(self.engine)
        return False


class _no_stream:
    def __enter__(self):
        return self

    def hand_back(self, *tensors):
        return tensors

    def __exit__(self, *exc):
        return False


def sharded_knn2(q, t_shard, train_base, *, ctx=None, group=None, local_knn=None, merge=None, gather=True):
    """Exact top-2 + ratio test with the train set sharded over the ranks of `group`.

    q        [nq, 128] uint8 tensor, identical on every rank
    t_shard  [nt_r, 128] uint8 tensor, this rank's contiguous shard of the train set
    train_base  global index of t_shard[0]
    local_knn(q, t, base) -> [nq, 4] int32 (d1, i1, d2, i2) and merge(parts[world, n, 4]) -> (out[n, 4], accept[n])
    default to the HIP kernels of `ctx`; tests inject CPU stand-ins to exercise the exchange under gloo.

    Returns (knn, accept, (lo, hi)).  With gather=True knn/accept cover all nq queries on every rank;
    otherwise they cover this rank's query slice [lo, hi).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # with the HIP kernels everything below runs on the engine's stream; injected stand-ins run wherever they like
    scope = _on_engine_stream(ctx) if (local_knn is None and merge is None and ctx is not None) else _no_stream()
    if local_knn is None:
        local_knn = _hip_local_knn(ctx)
        if not isinstance(scope, _on_engine_stream):
            local_knn = _synchronised(ctx, local_knn)
    if merge is None:
        merge = _hip_merge(ctx)
        if not isinstance(scope, _on_engine_stream):
            merge = _synchronised(ctx, merge)
    nq = q.shape[0]
    with scope:
        part = local_knn(q, t_shard, train_base)                       # [nq, 4]
        slice_len = (nq + world - 1) // world                          # equal slices, last one padded
        padded = torch.full((world * slice_len, 4), -1, dtype=torch.int32, device=part.device)
        padded[:, 0] = np.iinfo(np.int32).max
        padded[:, 2] = np.iinfo(np.int32).max
        padded[:nq] = part
        recv = torch.empty_like(padded)
        # slice s of every rank's partial result goes to rank s: 16 bytes per query per peer
        dist.all_to_all_single(recv, padded, group=group)
        merged, accept = merge(recv.view(world, slice_len, 4).contiguous())
        lo, hi = rank * slice_len, min(nq, (rank + 1) * slice_len)
        if not gather:
            n = max(0, hi - lo)
            merged, accept = scope.hand_back(merged[:n], accept[:n])
            return merged, accept, (lo, hi)
        all_m = torch.empty((world * slice_len, 4), dtype=torch.int32, device=merged.device)
        all_a = torch.empty((world * slice_len,), dtype=torch.uint8, device=merged.device)
        dist.all_gather_into_tensor(all_m, merged.contiguous(), group=group)
        dist.all_gather_into_tensor(all_a, accept.contiguous(), group=group)
        all_m, all_a = scope.hand_back(all_m[:nq], all_a[:nq])
        return all_m, all_a, (0, nq)
