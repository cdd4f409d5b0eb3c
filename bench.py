#!/usr/bin/env python3
"""Benchmark of the MI355X SIFT engine on the reference's headline workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = exec() (pyramid + DoG, extrema, clean, descriptors) over one batch
of synthetic 1920x1080 frames per GPU, frames already resident in HBM.  Frames
are distinct `dead-leaves` images (seed 1, frame index = global frame number),
sharded over ranks by frame index with no data-path collective ("weak"
scaling: every rank runs the same number of frames per step).

Rank 0 prints ONE JSON line: whole-job frames/s, the HBM roofline of the
dominant pyramid kernel measured with HIP events on the engine's own stream,
per-stage device time shares, the matcher's pairs/s and MFMA fraction, and the
CPU oracle timed on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_I8_PEAK_TFLOPS = 5000.0  # dense int8 MFMA peak = 2x the 2.5 PF bf16 dense peak


def pyramid_geometry(w, h):
    import sift_amd as S
    n = S.num_octaves(w, h)
    dims = [S.octave_dims(w, h, o) for o in range(n)]
    return dims


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pool", type=int, default=2, help="distinct batches resident in HBM, cycled over steps")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts (HIP streams) per GPU; the batch is split evenly between them")
    ap.add_argument("--knn-queries", type=int, default=262144)
    ap.add_argument("--knn-train", type=int, default=1 << 20)
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--force-sharded-knn", action="store_true",
                    help="run the sharded-matcher leg even with one rank (exercises the RCCL exchange path)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)

    import sift_amd as S

    W, H, B = args.width, args.height, args.batch
    nstreams = max(1, args.streams)
    if B % nstreams:
        raise SystemExit("--batch must be a multiple of --streams")
    Bs = B // nstreams
    ctxs = [S.Context(local_rank) for _ in range(nstreams)]
    batches = [c.batch(W, H, Bs) for c in ctxs]
    ctx, batch = ctxs[0], batches[0]

    # ---- synthetic frames, resident in HBM before the timed region -----------
    npool = max(1, args.pool)
    host = np.empty((npool * B, H, W), np.uint8)

    def make_frame(k):
        p, i = divmod(k, B)
        host[k] = S.synth_dead_leaves(1, (p * world + rank) * B + i, W, H)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 1) // max(1, world)))) as ex:
        list(ex.map(make_frame, range(npool * B)))  # the generator is a reentrant C function (GIL released)
    pool = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()

    def run_step(step):
        p = step % npool
        for k, bt in enumerate(batches):
            bt.set_external_frames(pool.data_ptr() + (p * B + k * Bs) * H * W)
            bt.run(Bs)

    def sync_all():
        for c in ctxs:
            c.synchronize()

    def barrier():
        sync_all()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for s in range(args.warmup):
        run_step(s)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(args.warmup + s)
    sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kp_mean = float(np.mean([bt.counts(Bs)["keypoints"].mean() for bt in batches]))
    frames_total = world * args.steps * B
    fps = frames_total / elapsed

    out = {
        "metric": "1080p frames/sec SIFT extract + 128-D desc pairs/sec KNN; % HBM/MFMA peak",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "batch of synthetic %dx%d dead-leaves frames (seed 1), image-sharded, "
                               "full exec(): pyramid+DoG, extrema, clean, descriptors" % (W, H),
                   "frames_per_gpu_per_step": B, "streams_per_gpu": nstreams, "distinct_batches_in_hbm": npool,
                   "keypoints_per_frame_mean": kp_mean},
    }

    # ---- sharded matcher: every rank holds knn_train rows of the train set and all queries;
    # local top-2, one all-to-all of 16 B per query per peer, merge on the owner of each slice
    if (world > 1 or args.force_sharded_knn) and not args.no_knn:
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        from sift_amd.dist import sharded_knn2
        nq, nt_r = args.knn_queries, args.knn_train
        t_shard = torch.from_numpy(S.synth_descriptors(5 + 97 * rank, nt_r)).cuda()
        q_all = torch.from_numpy(S.synth_queries(6, S.synth_descriptors(5, min(nt_r, 1 << 16)), nq)).cuda()
        sharded_knn2(q_all, t_shard, rank * nt_r, ctx=ctx, gather=False)      # warm-up (RCCL channels, kernels)
        torch.cuda.synchronize()
        dist.barrier()
        reps = 3
        t1 = time.perf_counter()
        for _ in range(reps):
            merged, accept, (lo, hi) = sharded_knn2(q_all, t_shard, rank * nt_r, ctx=ctx, gather=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # property check on this rank's slice: best distance <= second best, indices inside the global train set
        m = merged.cpu().numpy()
        ok = bool((m[:, 0] <= m[:, 2]).all() and (m[:, 1] >= 0).all() and (m[:, 1] < world * nt_r).all())
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        out["knn_sharded"] = {"queries": nq, "train_total": world * nt_r, "train_per_gpu": nt_r, "ms": dt * 1e3,
                              "pairs_per_s": nq * world * nt_r / dt, "exchange": "all_to_all_single, 16 B per query per peer",
                              "sane": bool(int(okt.item()))}
        del t_shard, q_all, merged, accept

    if rank == 0:
        # ---- per-kernel device time (HIP events on the engine's stream) ---------
        # (one context, one full-size batch, launches serialised by the event pairs)
        del batches[1:]
        if Bs != B:
            batches[0].close()
            batch = ctx.batch(W, H, B)
        batch.set_external_frames(pool.data_ptr())
        prof = None
        for _ in range(3):
            prof = batch.run_profiled(B)
        dims = pyramid_geometry(W, H)
        P = sum(w * h for w, h in dims)
        total_ms = sum(v["ms"] for v in prof.values())
        stages = {k: {"ms": round(v["ms"], 4), "launches": v["launches"],
                      "share": round(v["ms"] / total_ms, 4)} for k, v in prof.items()}
        out["stages_per_batch"] = stages
        # dominant pyramid kernel: the blur level with the largest device time.
        # Algorithmic bytes of one blur level over all octaves: read G[j-1],
        # write G[j], write D[j-1] = 3 * 8 B per pixel.
        blur = {k: v for k, v in prof.items() if k.startswith("blur_l")}
        dom = max(blur, key=lambda k: blur[k]["ms"])
        lv = blur[dom]
        bytes_per_launch = 3 * 8 * P * B / lv["launches"]
        avg_ms = lv["ms"] / lv["launches"]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        taps = {"blur_l1": 11, "blur_l2": 13, "blur_l3": 17, "blur_l4": 21, "blur_l5": 27}[dom]
        traffic = None
        tpath = os.path.join(HERE, "profiles", "traffic.json")
        if os.path.exists(tpath) and (W, H) == (1920, 1080):
            # measured with PMC counters on 1080p batches; bytes per launch scale with the frames per launch
            try:
                tj = json.load(open(tpath))
                traffic = tj[dom] * B / tj["_frames_per_launch"]
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "k_blur_march<%d> (%d taps)" % (taps // 2, taps),
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                           "launches_per_batch": lv["launches"]}
        pyr_ms = prof["base"]["ms"] + sum(v["ms"] for v in blur.values())
        pyr_bytes = (W * H + 8 * dims[0][0] * dims[0][1] + 15 * 8 * P) * B
        out["pyramid"] = {"ms_per_batch": pyr_ms, "algorithmic_GBs": pyr_bytes / (pyr_ms * 1e-3) / 1e9,
                          "frac_of_hbm_peak": pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

        # ---- matcher: pairs/s and MFMA fraction --------------------------------
        if not args.no_knn:
            nq, nt = args.knn_queries, args.knn_train
            tr = S.synth_descriptors(5, nt)
            qq = S.synth_queries(6, tr, nq)
            d_t = torch.from_numpy(tr).cuda()
            d_q = torch.from_numpy(qq).cuda()
            d_o = torch.empty((nq, 4), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ctx.knn2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 0, d_o.data_ptr())
            ctx.synchronize()
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                ctx.knn2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 0, d_o.data_ptr())
            ctx.synchronize()
            dt = (time.perf_counter() - t1) / reps
            pairs = nq * nt / dt
            tflops = 2 * 128 * nq * nt / dt / 1e12
            out["knn"] = {"queries": nq, "train": nt, "ms": dt * 1e3, "pairs_per_s": pairs,
                          "roofline": {"bound": "mfma", "achieved": tflops, "peak": MFMA_I8_PEAK_TFLOPS,
                                       "unit": "TOP/s (int8)", "frac": tflops / MFMA_I8_PEAK_TFLOPS}}
            del d_t, d_q, d_o

        # ---- CPU oracle on a bounded sample of the same workload ----------------
        if not args.no_cpu and world == 1:  # the CPU baseline is reported at N = 1 only
            from oracle import oracle as O
            cores = min(16, os.cpu_count() or 1)
            tcpu = time.perf_counter()
            stage = {}
            for i in range(args.cpu_frames):
                o = O.OracleSift(host[i])
                o.exec(threads=cores, blur_threads=cores)
                for k, v in o.timings.items():
                    stage[k] = stage.get(k, 0.0) + v
                o.close()
            tcpu = time.perf_counter() - tcpu
            out["cpu_baseline"] = {"value": args.cpu_frames / tcpu, "unit": "frames/s", "cores": cores,
                                   "kind": "port",
                                   "sample": "%d of the same %dx%d frames through the CPU oracle (all stages, "
                                             "%d threads incl. blur rows)" % (args.cpu_frames, W, H, cores),
                                   "stage_seconds": {k: round(v, 4) for k, v in stage.items()}}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
