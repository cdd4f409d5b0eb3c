#!/usr/bin/env python3
"""Benchmark of the MI355X SIFT engine on the reference's headline workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = exec() (pyramid, extrema + gradient planes, localisation and
orientation, clean, descriptors) over one batch of synthetic 1920x1080 frames
per GPU, frames already resident in HBM.  Frames are distinct `dead-leaves`
images (seed 1, frame index = global frame number), sharded over ranks by frame
index with no data-path collective ("weak" scaling: every rank runs the same
number of frames per step).

Rank 0 prints ONE JSON line: whole-job frames/s; the HBM roofline of the
dominant pyramid kernel measured with HIP events on the engine's own stream,
and a bound + fraction for every other heavy kernel; per-stage device times;
host-frames-in / host-results-out throughput; the matcher on the per-GPU share
of the 1 M x 10 M configuration (search + merge + ratio test) with its MFMA
fraction; the CPU oracle timed on bounded samples of both workloads; and, at
N = 1, a `configs` block with the 3840x2160 batch and the single 2048x1080
frame of the other benchmark configurations.

Launched under torchrun (WORLD_SIZE set) every rank drives one GPU and the
matcher's exchange is torch.distributed's all-to-all.  Launched as one plain
process with --gpus N > 1, that process drives all N GPUs itself through the C
group (sift_group_*): resident batches on every GPU for `value`, then
sift_extract_batch_u8 and sift_knn2_sharded over the group (`c_group`).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak
MFMA_I8_PEAK_TFLOPS = 5000.0  # dense int8 MFMA peak = 2x the 2.5 PF bf16 dense peak
# ds_add_u64 retires one wave instruction (up to 64 lanes) per 7.2 clocks per CU when no two lanes of a 16-lane
# pass collide (scripts/ubench/lds_atomic3.hip): 256 CUs x 64 / 7.2 lanes/clk x 2.4 GHz
LDS_U64_ATOMIC_PEAK_GOPS = 256 * 64 / 7.2 * 2.4


def source_hash():
    """Identity of the kernel sources (device code and the headers it includes; the host-side .cpp files do not
    change what a kernel moves): counter-derived figures in profiles/ are only quoted for the kernels they were
    measured on."""
    h = hashlib.sha256()
    d = os.path.join(HERE, "sift_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) and name != "abi_internal.h":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pyramid_geometry(w, h):
    import sift_amd as S
    return [S.octave_dims(w, h, o) for o in range(S.num_octaves(w, h))]


def kernel_report(prof, W, H, B, traffic, work=None):
    """roofline / pyramid / kernels blocks from one profiled pass (per-stage device times) of B WxH frames."""
    dims = pyramid_geometry(W, H)
    P = sum(w * h for w, h in dims)
    P0 = dims[0][0] * dims[0][1]
    P_scanned = sum(w * h for w, h in dims if w > 10 and h > 10)   # octaves with a non-empty scan range
    total_ms = sum(v["ms"] for v in prof.values())
    rep = {"stages_per_batch": {k: {"ms": round(v["ms"], 4), "launches": v["launches"], "share": round(v["ms"] / total_ms, 4)}
                                for k, v in prof.items() if v["launches"]}}
    # Pyramid launches.  A blur level is one launch per octave (blur_l1..blur_l5); in the batch schedule levels 1 and 2
    # of the large octaves are one launch that does not read level 1 back (blur_l12).  Algorithmic bytes of a blur
    # level, as the reference formulates the work (GaussianBlur, sift.cpp:150, then subtract, sift.cpp:173): read
    # G[j-1], write G[j], write D[j-1] = 3 x 8 B per pixel.  This engine never stores D (its two readers subtract in
    # registers), so the bytes a level has to MOVE are 2 x 8 B per pixel, and 3 x 8 B for the two fused levels
    # together; those figures are reported beside the algorithmic ones as `moved`.
    blur = {k: v for k, v in prof.items() if k.startswith("blur_l") and v["launches"]}
    n_pair = prof.get("blur_l12", {"launches": 0})["launches"]          # the largest octaves
    P_pair = sum(w * h for w, h in dims[:n_pair])

    def level_px(k):   # pixels a stage's launches cover, and the levels each of them produces
        if k == "blur_l12":
            return P_pair, 2
        return (P - P_pair if k in ("blur_l1", "blur_l2") else P), 1

    # Dominant pyramid kernel: the launch kind with the largest device time per pyramid level it produces.
    dom = max(blur, key=lambda k: blur[k]["ms"] / level_px(k)[1])
    lv = blur[dom]
    px, nlev = level_px(dom)
    bytes_per_launch = nlev * 3 * 8 * px * B / lv["launches"]
    moved_per_launch = (nlev + 1) * 8 * px * B / lv["launches"]
    avg_ms = lv["ms"] / lv["launches"]
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    moved = moved_per_launch / (avg_ms * 1e-3) / 1e9
    kname = {"blur_l1": "k_blur_march<5> (11 taps)", "blur_l2": "k_blur_march<6> (13 taps)", "blur_l3": "k_blur_march<8> (17 taps)",
             "blur_l4": "k_blur_march<10> (21 taps)", "blur_l5": "k_blur_march<13> (27 taps)",
             "blur_l12": "k_blur_pair<5, 6> (11 + 13 taps)"}[dom]
    rep["roofline"] = {"bound": "hbm", "kernel": kname,
                       "selection": "the pyramid launch kind with the largest device time per level it produces",
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS,
                       "traffic": traffic.get(dom),   # mean HBM bytes per launch, like the algorithmic figure
                       "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                       "launches_per_batch": lv["launches"],
                       "moved": {"bytes_per_launch": moved_per_launch, "achieved": moved, "frac": moved / HBM_PEAK_GBS,
                                 "note": "bytes this kernel has to move: the difference-of-Gaussians level is not stored"}}
    pyr_ms = prof["base"]["ms"] + sum(v["ms"] for v in blur.values())
    pyr_bytes = (W * H + 8 * P0 + 15 * 8 * P) * B
    pyr_moved = (W * H + 8 * P0 + 10 * 8 * P - 8 * P_pair) * B
    rep["pyramid"] = {"ms_per_batch": pyr_ms, "algorithmic_GBs": pyr_bytes / (pyr_ms * 1e-3) / 1e9,
                      "frac_of_hbm_peak": pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "moved_bytes_per_frame": pyr_moved / B,
                      "moved_GBs": pyr_moved / (pyr_ms * 1e-3) / 1e9,
                      "moved_frac_of_hbm_peak": pyr_moved / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "per_level_moved_frac": {k: round((level_px(k)[1] + 1) * 8 * level_px(k)[0] * B / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                               for k, v in sorted(blur.items()) if level_px(k)[0] * 50 > P}}
    # every other heavy kernel: what bounds it and how close it runs to that bound
    kernels = {}
    ex = prof["extrema"]
    ex_bytes = (6 * 8 + 3 * 16) * P_scanned * B      # six Gaussian levels read, three gradient planes written
    kernels["extrema_gradients"] = {
        "kernel": "k_extrema_grad", "bound": "hbm", "ms_per_batch": ex["ms"],
        "algorithmic_bytes_per_batch": ex_bytes, "achieved": ex_bytes / (ex["ms"] * 1e-3) / 1e9, "unit": "GB/s",
        "peak": HBM_PEAK_GBS, "frac": ex_bytes / (ex["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": traffic["k_extrema_grad"] * ex["launches"] if "k_extrema_grad" in traffic else None}
    if work is not None:
        dsc = prof["descriptor"]
        upd = float(work[:, 1].sum())
        kernels["descriptor"] = {
            "kernel": "k_descriptors", "bound": "vector ALU issue beside the LDS atomic pipe", "ms_per_batch": dsc["ms"],
            "window_samples_per_batch": float(work[:, 0].sum()), "histogram_updates_per_batch": upd,
            "achieved": upd / (dsc["ms"] * 1e-3) / 1e9, "unit": "G ds_add_u64 lanes/s",
            "peak": LDS_U64_ATOMIC_PEAK_GOPS, "frac": upd / (dsc["ms"] * 1e-3) / 1e9 / LDS_U64_ATOMIC_PEAK_GOPS,
            "note": "peak = conflict-free ds_add_u64 rate with every lane active; by counters (profiles/) the vector "
                    "ALUs are busy 74 % and the LDS 29 % of the kernel's cycles",
            "traffic": traffic.get("k_descriptors")}
    base = prof["base"]
    base_bytes = (W * H + 8 * P0) * B
    kernels["base"] = {"kernel": "k_base_march<5>", "bound": "hbm (write-only stream)", "ms_per_batch": base["ms"],
                       "achieved": base_bytes / (base["ms"] * 1e-3) / 1e9, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                       "frac": base_bytes / (base["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic.get("base")}
    rep["kernels"] = kernels
    return rep


def config_4k(S, torch, ctx, frames_per_step, steps):
    """BASELINE config 4: 3840x2160 dead-leaves frames (seed 3), 11 octaves, 7680x4320 base."""
    W, H, B = 3840, 2160, frames_per_step
    host = np.empty((B, H, W), np.uint8)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(lambda i: host.__setitem__(i, S.synth_dead_leaves(3, i, W, H)), range(B)))
    pool = torch.from_numpy(host).cuda()
    del host
    batch = ctx.batch(W, H, B)
    batch.set_external_frames(pool.data_ptr())
    batch.run(B)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(B)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    kp = float(batch.counts(B)["keypoints"].mean())
    prof = None
    for _ in range(2):
        prof = batch.run_profiled(B)
    rep = kernel_report(prof, W, H, B, {})
    batch.close()
    del pool
    torch.cuda.empty_cache()
    out = {"workload": "%d synthetic 3840x2160 dead-leaves frames (seed 3) per step, resident in HBM, full exec()" % B,
           "frames_per_s": steps * B / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "frames_per_step": B, "octaves": S.num_octaves(W, H), "keypoints_per_frame_mean": kp}
    out.update(rep)
    return out


def config_single_frame(S, ctx, with_cpu):
    """BASELINE config 2: the reference's res/2k.png is absent from its checkout; one 2048x1080 dead-leaves frame
    (seed 2) stands in for it.  The whole pipeline on one frame, host buffer in, host keypoints and descriptors out."""
    W, H = 2048, 1080
    frame = S.synth_dead_leaves(2, 0, W, H)
    for _ in range(3):
        kps, desc, info = ctx.extract(frame)
    times = []
    for _ in range(20):
        t0 = time.perf_counter()
        ctx.extract(frame)
        times.append(time.perf_counter() - t0)
    _, _, ms = ctx.extract_stages(frame)
    names = ["gen_gaussian_images", "gen_dog_images", "gen_scale_space_extrema", "clean_keypoints", "get_descriptors"]
    out = {"workload": "one 2048x1080 dead-leaves frame (seed 2) standing in for the reference's absent res/2k.png: "
                       "sift_extract_gray8, host frame in, host keypoints + descriptors out",
           "host_to_host_ms_median": 1e3 * float(np.median(times)), "host_to_host_ms_min": 1e3 * float(np.min(times)),
           "calls": len(times), "keypoints": int(info["keypoints"]), "candidates": int(info["candidates"]),
           "stage_ms": {k: round(v, 4) for k, v in zip(names, ms)},
           "stage_note": "device time between events at the reference's TIMEIT boundaries (sift.cpp:71-92) in the "
                         "three-stream schedule of a single frame: levels 4-5 of an octave run beside the next "
                         "octave's levels 1-3, so the first two boundaries are approximate; gen_dog_images is 0 "
                         "(no such pass)"}
    if with_cpu:
        from oracle import oracle as O
        cores = min(16, os.cpu_count() or 1)
        o = O.OracleSift(frame)
        t0 = time.perf_counter()
        o.exec(threads=cores, blur_threads=cores)
        dtc = time.perf_counter() - t0
        n_cpu = len(o.keypoints())
        stage = {k: round(v, 4) for k, v in o.timings.items()}
        o.close()
        out["cpu_baseline"] = {"value": dtc * 1e3, "unit": "ms per frame", "cores": cores, "kind": "port",
                               "sample": "the same frame through the CPU oracle, all stages", "stage_seconds": stage,
                               "keypoints": n_cpu, "gpu_keypoint_count_equal": bool(n_cpu == int(info["keypoints"]))}
    return out


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
    ap.add_argument("--knn-queries", type=int, default=1 << 20, help="queries (the 1 M of the 1 M x 10 M configuration)")
    ap.add_argument("--knn-train", type=int, default=1310720, help="train rows per GPU (10 M / 8)")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--force-sharded-knn", action="store_true",
                    help="run the sharded-matcher leg even with one rank (exercises the RCCL exchange path)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--cpu-knn-seconds", type=float, default=10.0)
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-frames", type=int, default=1024)
    ap.add_argument("--no-configs", action="store_true", help="skip the 3840x2160 and single-frame legs")
    ap.add_argument("--frames-4k", type=int, default=16)
    ap.add_argument("--steps-4k", type=int, default=3)
    ap.add_argument("--share-device", action="store_true",
                    help="single-process --gpus N: put every member on GPU 0 (rehearsal of the group path on a one-GPU box)")
    args = ap.parse_args()

    # The contract is ONE line on stdout.  Libraries loaded below write there too (RCCL prints a version banner when
    # a communicator is created), so for the length of the run descriptor 1 points at stderr, and the line goes to the
    # real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the several-rank flow on a box with one GPU (RCCL takes one rank per device): every rank on GPU 0
    # and the process group on gloo.  SIFT_BENCH_REHEARSAL=1 torchrun ... bench.py --gpus N --no-knn --batch 8
    rehearsal = os.environ.get("SIFT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)

    import sift_amd as S

    # one plain process asked for several GPUs: it drives them all (the C group); under torchrun a rank drives one
    members = args.gpus if (world == 1 and "WORLD_SIZE" not in os.environ and args.gpus > 1) else 1
    devices = [0 if args.share_device else d for d in range(members)] if members > 1 else [local_rank]
    if members > 1 and not args.share_device and S.device_count() < members:
        raise SystemExit("--gpus %d in one process needs that many GPUs (%d visible)" % (members, S.device_count()))
    parts = world * members   # GPUs (or members) over which the frames are sharded

    W, H, B = args.width, args.height, args.batch
    nstreams = max(1, args.streams) if members == 1 else 1
    if B % nstreams:
        raise SystemExit("--batch must be a multiple of --streams")
    Bs = B // nstreams
    ctxs = [S.Context(d) for d in devices for _ in range(nstreams)]
    batches = [c.batch(W, H, Bs) for c in ctxs]
    ctx, batch = ctxs[0], batches[0]

    # ---- synthetic frames, resident in HBM before the timed region -----------
    npool = max(1, args.pool)
    host = np.empty((members, npool * B, H, W), np.uint8)

    def make_frame(mk):
        m, k = mk
        p, i = divmod(k, B)
        host[m, k] = S.synth_dead_leaves(1, (p * parts + rank * members + m) * B + i, W, H)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 1) // max(1, world)))) as ex:
        list(ex.map(make_frame, [(m, k) for m in range(members) for k in range(npool * B)]))  # reentrant C generator (GIL released)
    pools = [torch.from_numpy(host[m]).to("cuda:%d" % devices[m]) for m in range(members)]
    pool = pools[0]
    host = host[0]
    for d in set(devices):
        torch.cuda.synchronize(d)

    def run_step(step):
        p = step % npool
        for k, bt in enumerate(batches):
            m, sub = divmod(k, nstreams)
            bt.set_external_frames(pools[m].data_ptr() + (p * B + sub * Bs) * H * W)
            bt.run(Bs)   # enqueues and returns: one host thread keeps every GPU fed

    def sync_all():
        for c in ctxs:
            c.synchronize()

    def barrier():
        sync_all()
        for d in set(devices):
            torch.cuda.synchronize(d)
        if dist is not None:
            dist.barrier()

    for s in range(args.warmup):
        run_step(s)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(args.warmup + s)
    sync_all()
    for d in set(devices):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kp_mean = float(np.mean([bt.counts(Bs)["keypoints"].mean() for bt in batches]))
    frames_total = parts * args.steps * B
    fps = frames_total / elapsed

    out = {
        "metric": "1080p frames/sec SIFT extract + 128-D desc pairs/sec KNN; % HBM/MFMA peak",
        "value": fps, "unit": "frames/s", "n_gpus": parts, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "batch of synthetic %dx%d dead-leaves frames (seed 1), image-sharded, "
                               "full exec(): pyramid, extrema + gradients, localise + orient, clean, descriptors" % (W, H),
                   "frames_per_gpu_per_step": B, "streams_per_gpu": nstreams, "distinct_batches_in_hbm": npool,
                   "keypoints_per_frame_mean": kp_mean,
                   "driver": ("one process per GPU (torchrun)" if world > 1 else
                              "one process driving %d GPUs through the C group%s" % (members, " (all on GPU 0)" if args.share_device else "")
                              if members > 1 else "one process, one GPU")},
    }

    # ---- one process, several GPUs: the C group's two entry points over all members ----------
    if members > 1:
        for bt in batches[1:]:
            bt.close()
        del batches[1:]
        grp = S.Group(ctxs)
        cg = {"members": members, "devices": devices}
        if not args.no_e2e:
            nf = members * max(B, min(args.e2e_frames, 512) // B * B)   # 1 GB of host frames per member
            e2e_host = np.concatenate([host] * ((nf + host.shape[0] - 1) // host.shape[0]))[:nf]
            grp.extract_batch_raw(e2e_host[:members * B], B)       # workspaces, page-locked buffers
            t1 = time.perf_counter()
            total_kp = grp.extract_batch_raw(e2e_host, -B)
            dt = time.perf_counter() - t1
            cg["extract_batch"] = {"frames": nf, "frames_per_s": nf / dt, "keypoints": total_kp,
                                   "what": "sift_extract_batch_u8 over the group: contiguous blocks of host frames per "
                                           "member, a host thread and a double-buffered pipeline each, no collective"}
            del e2e_host
        if not args.no_knn:
            nq, nt = args.knn_queries, args.knn_train * members
            tr = S.synth_descriptors(5, nt)
            qq = S.synth_queries(6, tr, nq)
            grp.knn2(qq[:4096], tr[:65536])
            t1 = time.perf_counter()
            knn, acc = grp.knn2(qq, tr)
            dt = time.perf_counter() - t1
            ok = bool((knn["d1"] <= knn["d2"]).all() and (knn["i1"] >= 0).all() and (knn["i1"] < nt).all())
            cg["knn2_sharded"] = {"queries": nq, "train_total": nt, "ms": dt * 1e3, "pairs_per_s": nq * nt / dt,
                                  "accepted": int(acc.sum()), "sane": ok,
                                  "what": "sift_knn2_sharded: host operands in, host results out (uploads included), "
                                          "train rows sharded over the members, 16 B per query per peer exchanged, merge"}
            del tr, qq
        grp.close()
        out["c_group"] = cg

    # ---- sharded matcher: every rank holds knn_train rows of the train set and all queries;
    # local top-2, one all-to-all of 16 B per query per peer, merge on the owner of each slice
    if (world > 1 or args.force_sharded_knn) and not args.no_knn and not rehearsal:  # gloo has no GPU all-to-all
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
        for bt in batches[1:]:
            bt.close()
        del batches[1:]
        if Bs != B:
            batches[0].close()
            batch = ctx.batch(W, H, B)
        batch.set_external_frames(pool.data_ptr())
        prof = None
        for _ in range(3):
            prof = batch.run_profiled(B)

        traffic = {}
        tpath = os.path.join(HERE, "profiles", "traffic.json")
        if os.path.exists(tpath) and (W, H) == (1920, 1080):
            # measured with PMC counters on 1080p batches; only quoted for the kernel sources they were measured on
            try:
                tj = json.load(open(tpath))
                if tj.get("_source_hash") == source_hash():
                    traffic = {k: v * B / tj["_frames_per_launch"] for k, v in tj.items()
                               if not k.startswith("_") and not k.endswith("_detail")}
            except Exception:
                traffic = {}
        out.update(kernel_report(prof, W, H, B, traffic, batch.descriptor_work(B)))
        batch.close()
        batch = None

        # ---- host frames in -> host keypoints + descriptors out ------------------
        if not args.no_e2e:
            g = S.Group([ctx])
            # launches of at most the resident batches' size, cut by the library: a quarter-size launch first and last
            nf = max(B, args.e2e_frames // B * B)
            e2e_host = np.concatenate([host] * ((nf + host.shape[0] - 1) // host.shape[0]))[:nf]
            g.extract_batch_raw(e2e_host[:2 * B], B)         # workspaces, page-locked buffers
            t1 = time.perf_counter()
            total_kp = g.extract_batch_raw(e2e_host, -B)
            dt = time.perf_counter() - t1
            out["end_to_end"] = {"frames": nf, "frames_per_s": nf / dt, "ms_per_frame": 1e3 * dt / nf,
                                 "keypoints": total_kp,
                                 "what": "sift_extract_batch_u8: pageable host frames in, host keypoints + 128-byte "
                                         "descriptors out, at most %d frames per launch with a quarter-size launch first "
                                         "and last, uploads and downloads through page-locked staging overlapped with "
                                         "the kernels of the neighbouring chunks" % B}
            del e2e_host
            g.close()

        # ---- the other benchmark configurations (N = 1 only, like the CPU baseline) ----------
        if not args.no_configs and parts == 1:
            del pools[:]
            pool = None
            torch.cuda.empty_cache()
            out["configs"] = {"3840x2160_batch": config_4k(S, torch, ctx, args.frames_4k, args.steps_4k),
                              "single_2048x1080_frame": config_single_frame(S, ctx, not args.no_cpu)}

        # ---- matcher: pairs/s and MFMA fraction on the per-GPU share of 1 M x 10 M ------
        if not args.no_knn:
            nq, nt = args.knn_queries, args.knn_train
            tr = S.synth_descriptors(5, nt)
            qq = S.synth_queries(6, tr, nq)
            d_t = torch.from_numpy(tr).cuda()
            d_q = torch.from_numpy(qq).cuda()
            d_p = torch.empty((nq, 4), dtype=torch.int32, device="cuda")
            d_o = torch.empty((nq, 4), dtype=torch.int32, device="cuda")
            d_a = torch.empty((nq,), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()

            def match():
                ctx.knn2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 0, d_p.data_ptr())
                ctx.knn2_merge_device(d_p.data_ptr(), 1, nq, d_o.data_ptr(), d_a.data_ptr())   # ratio test

            match()
            ctx.synchronize()
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                match()
            ctx.synchronize()
            dt = (time.perf_counter() - t1) / reps
            pairs = nq * nt / dt
            tflops = 2 * 128 * nq * nt / dt / 1e12
            out["knn"] = {"queries": nq, "train": nt, "ms": dt * 1e3, "pairs_per_s": pairs,
                          "accepted": int(d_a.sum().item()),
                          "what": "sift_knn2_device + sift_knn2_merge_device with the ratio test, operands in HBM; "
                                  "one GPU's share of 1 M queries x 10 M train rows sharded 8 ways",
                          "roofline": {"bound": "mfma", "achieved": tflops, "peak": MFMA_I8_PEAK_TFLOPS,
                                       "unit": "TOP/s (int8)", "frac": tflops / MFMA_I8_PEAK_TFLOPS}}
            if not args.no_cpu and world == 1:
                from oracle import oracle as O
                cores = min(16, os.cpu_count() or 1)
                # bounded sample: as many of the same queries against the same train rows as fit the time budget
                probe = 16
                t1 = time.perf_counter()
                O.knn2_u8(qq[:probe], tr, threads=cores)
                rate = probe * nt / (time.perf_counter() - t1)
                n_s = int(max(probe, min(nq, args.cpu_knn_seconds * rate / nt)))
                t1 = time.perf_counter()
                ci1, cd1, ci2, cd2, cacc = O.knn2_u8(qq[:n_s], tr, threads=cores)
                dtc = time.perf_counter() - t1
                got = d_o[:n_s].cpu().numpy()
                same = bool(np.array_equal(got[:, 0], cd1) and np.array_equal(got[:, 1], ci1) and
                            np.array_equal(got[:, 2], cd2) and np.array_equal(got[:, 3], ci2) and
                            np.array_equal(d_a[:n_s].cpu().numpy(), cacc))
                out["knn"]["cpu_baseline"] = {"value": n_s * nt / dtc, "unit": "pairs/s", "cores": cores, "kind": "port",
                                              "sample": "the first %d of the same queries against all %d train rows "
                                                        "through the CPU oracle's exact brute force" % (n_s, nt),
                                              "gpu_results_identical_on_sample": same}
            del d_t, d_q, d_p, d_o, d_a

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
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
