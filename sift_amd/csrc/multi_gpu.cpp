// The two multi-GPU splits of the engine behind the C ABI, for a C++ caller that owns all GPUs of a node from
// one process (the reference's caller is a single C++ program, src/main.cpp:48-75):
//
//  * extraction: frames are independent, so a run of host frames is cut into contiguous blocks, one block per
//    GPU, with no data-path collective.  Every GPU is driven by its own host thread through a double-buffered
//    pipeline (upload of chunk k+1 and download of chunk k-1 overlap the kernels of chunk k);
//  * matching: the train descriptors are cut into contiguous shards, every GPU holds all queries and finds its
//    local top-2 with global train indices; then each GPU sends, for every query slice, its partial results to
//    the GPU that owns the slice (grouped ncclSend/ncclRecv, 16 bytes per query per peer, over xGMI), which
//    merges them under (distance, index) order and applies the ratio test.  The result does not depend on the
//    number of GPUs.  A group with two members on one GPU cannot use RCCL (one communicator rank per device):
//    its slices travel as device-to-device copies on the sender's stream, ordered by events.
//
// RCCL is resolved at run time (dlopen of librccl.so.1) on the first sharded matcher call that needs it, so that
// the library has no load-time dependency on it and shares whichever copy the process already holds.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "abi_internal.h"

using namespace sa;

namespace {

size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int load_rccl(Rccl &r) {
    if (r.lib) return SIFT_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(SIFT_ERR_HIP, std::string("cannot load RCCL: ") + dlerror());
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) ok = false;
        return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(h);
        return fail(SIFT_ERR_HIP, "librccl lacks an expected entry point");
    }
    r.lib = h;
    return SIFT_OK;
}

// Double-buffered extraction pipeline of one GPU.
struct Pipe {
    int w = 0, h = 0, chunk = 0, cap = 0;
    sift_batch *ws[2] = {nullptr, nullptr};
    void *pin_up[2] = {nullptr, nullptr};
    void *pin_down = nullptr;
    size_t pin_down_bytes = 0;
    hipStream_t up = nullptr, down = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr};
};

struct DevBuffers {
    void *buf = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct sift_group {
    std::vector<sift_ctx *> ctx;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<DevBuffers> knn;
    std::vector<Pipe> pipe;
    std::vector<hipEvent_t> sent;  // copy transport: member r's slices have all been copied out
};

namespace {

void pipe_release(sift_ctx *ctx, Pipe &p) {
    hipSetDevice(ctx->device);
    for (int i = 0; i < 2; i++) {
        if (p.ws[i]) sift_batch_destroy(p.ws[i]);
        if (p.pin_up[i]) hipHostFree(p.pin_up[i]);
        if (p.ev_up[i]) hipEventDestroy(p.ev_up[i]);
        if (p.ev_run[i]) hipEventDestroy(p.ev_run[i]);
    }
    if (p.pin_down) hipHostFree(p.pin_down);
    if (p.up) hipStreamDestroy(p.up);
    if (p.down) hipStreamDestroy(p.down);
    p = Pipe();
}

int pipe_prepare(sift_ctx *ctx, Pipe &p, int w, int h, int chunk) {
    if (p.ws[0] && p.w == w && p.h == h && p.chunk == chunk && p.cap == ctx->single_cap) return SIFT_OK;
    pipe_release(ctx, p);
    HIP_TRY(hipSetDevice(ctx->device));
    p.w = w;
    p.h = h;
    p.chunk = chunk;
    p.cap = ctx->single_cap;
    for (int i = 0; i < 2; i++) {
        int rc = sift_batch_create(ctx, w, h, chunk, p.cap, &p.ws[i]);
        if (rc) return rc;
        HIP_TRY(hipHostMalloc(&p.pin_up[i], (size_t)chunk * w * h, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&p.ev_up[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p.ev_run[i], hipEventDisableTiming));
    }
    HIP_TRY(hipStreamCreateWithFlags(&p.up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p.down, hipStreamNonBlocking));
    return SIFT_OK;
}

struct ExtractJob {
    const uint8_t *frames;
    size_t row_stride, frame_stride;
    int lo, hi;  // this GPU's frames [lo, hi)
    sift_frame_result *results;
    bool ramp;   // the library chooses the launch sizes (the caller named none)
    int launch;  // frames per launch: at most the workspace's capacity, which an earlier call may have left larger
};

// Results of chunk `c` (frames first .. first + m) to the host: counts, then keypoints and descriptors of every
// frame in one pass through page-locked memory.  A frame that overflowed its lists is run again through the
// single-image call, which grows its workspace (the reference's vectors are unbounded, sift.hpp:93-94).
int pipe_download(sift_ctx *ctx, Pipe &p, int slot, const ExtractJob &job, int first, int m) {
    sift_batch *b = p.ws[slot];
    HIP_TRY(hipEventSynchronize(p.ev_run[slot]));
    std::vector<FrameCounters> fc((size_t)m);
    HIP_TRY(hipMemcpyAsync(fc.data(), b->d_counters, (size_t)m * sizeof(FrameCounters), hipMemcpyDeviceToHost, p.down));
    HIP_TRY(hipStreamSynchronize(p.down));
    size_t total = 0;
    std::vector<size_t> off((size_t)m);
    std::vector<char> redo((size_t)m, 0);
    for (int i = 0; i < m; i++) {
        redo[i] = fc[i].overflow || fc[i].candidates > b->cand_cap || fc[i].oriented > b->kp_cap;
        off[i] = total;
        if (!redo[i]) total += up256((size_t)fc[i].keypoints * sizeof(DevKeypoint)) + up256((size_t)fc[i].keypoints * 128);
    }
    if (total > p.pin_down_bytes) {
        if (p.pin_down) HIP_TRY(hipHostFree(p.pin_down));
        p.pin_down = nullptr;
        p.pin_down_bytes = 0;
        const size_t want = (total * 3 / 2 + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        HIP_TRY(hipHostMalloc(&p.pin_down, want, hipHostMallocDefault));
        p.pin_down_bytes = want;
    }
    char *stage = static_cast<char *>(p.pin_down);
    for (int i = 0; i < m; i++) {
        const size_t n = (size_t)fc[i].keypoints;
        if (redo[i] || n == 0) continue;
        HIP_TRY(hipMemcpyAsync(stage + off[i], b->d_final + (size_t)i * b->kp_cap, n * sizeof(DevKeypoint), hipMemcpyDeviceToHost,
                               p.down));
        HIP_TRY(hipMemcpyAsync(stage + off[i] + up256(n * sizeof(DevKeypoint)), b->d_desc + (size_t)i * b->kp_cap * 128, n * 128,
                               hipMemcpyDeviceToHost, p.down));
    }
    HIP_TRY(hipStreamSynchronize(p.down));
    for (int i = 0; i < m; i++) {
        sift_frame_result &r = job.results[first + i];
        if (redo[i]) {
            sift_result one;
            const uint8_t *src = job.frames + (size_t)(first + i) * job.frame_stride;
            int rc = sift_extract_gray8(ctx, src, p.w, p.h, job.row_stride, &one);
            if (rc) return rc;
            r.n = one.n;
            r.counts = one.counts;
            r.keypoints = one.keypoints;  // ownership moves to the frame result
            r.descriptors = one.n ? (uint8_t *)std::malloc((size_t)one.n * 128) : nullptr;
            if (one.n && !r.descriptors) {
                std::free(one.descriptors);
                return fail(SIFT_ERR_INVALID, "out of host memory");
            }
            for (size_t k = 0; k < (size_t)one.n * 128; k++) r.descriptors[k] = (uint8_t)one.descriptors[k];
            std::free(one.descriptors);
            continue;
        }
        const size_t n = (size_t)fc[i].keypoints;
        static_assert(sizeof(sift_frame_counts) == sizeof(FrameCounters), "layout");
        std::memcpy(&r.counts, &fc[i], sizeof(FrameCounters));
        r.n = (int32_t)n;
        if (n == 0) continue;
        r.keypoints = (sift_keypoint *)std::malloc(n * sizeof(sift_keypoint));
        r.descriptors = (uint8_t *)std::malloc(n * 128);
        if (!r.keypoints || !r.descriptors) return fail(SIFT_ERR_INVALID, "out of host memory");
        const DevKeypoint *src = reinterpret_cast<const DevKeypoint *>(stage + off[i]);
        for (size_t k = 0; k < n; k++) std::memcpy(&r.keypoints[k], &src[k], sizeof(sift_keypoint));
        std::memcpy(r.descriptors, stage + off[i] + up256(n * sizeof(DevKeypoint)), n * 128);
    }
    return SIFT_OK;
}

// Launch sizes for `total` frames through workspaces of `cap` frames.  When the library chooses: a quarter-size launch first and last, so that the
// GPU starts after a quarter of the staging and upload time of a full launch and the last download and unpacking,
// which nothing overlaps, is a quarter as long; full or evenly cut launches between them.
std::vector<int> launch_sizes(int total, int cap, bool ramp) {
    std::vector<int> sizes;
    if (total <= cap) {
        if (total > 0) sizes.push_back(total);
        return sizes;
    }
    if (!ramp) {  // the caller's launch size, the remainder last
        for (int at = 0; at < total; at += cap) sizes.push_back(std::min(cap, total - at));
        return sizes;
    }
    const int small = std::max(1, cap / 4);
    int rest = total - small;
    const int tail = rest > cap ? small : 0;
    rest -= tail;
    const int n = (rest + cap - 1) / cap;
    sizes.push_back(small);
    for (int i = 0; i < n; i++) sizes.push_back(rest / n + (i < rest % n ? 1 : 0));
    if (tail) sizes.push_back(tail);
    return sizes;
}

int pipe_run(sift_ctx *ctx, Pipe &p, const ExtractJob &job) {
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t px = (size_t)p.w * p.h;
    const std::vector<int> sizes = launch_sizes(job.hi - job.lo, std::min(job.launch, p.chunk), job.ramp);
    const int nchunks = (int)sizes.size();
    std::vector<int> firsts((size_t)nchunks);
    for (int k = 0, at = job.lo; k < nchunks; at += sizes[k], k++) firsts[k] = at;
    for (int k = 0; k < nchunks; k++) {
        const int slot = k & 1;
        const int first = firsts[k], m = sizes[k];
        sift_batch *b = p.ws[slot];
        // page-locked staging of chunk k (the transfer of chunk k-2 out of this buffer has completed: its run was
        // waited for when its results were taken)
        char *stage = static_cast<char *>(p.pin_up[slot]);
        auto stage_frames = [&](int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                const uint8_t *src = job.frames + (size_t)(first + i) * job.frame_stride;
                if (job.row_stride == (size_t)p.w) {
                    std::memcpy(stage + (size_t)i * px, src, px);
                } else {
                    for (int y = 0; y < p.h; y++)
                        std::memcpy(stage + (size_t)i * px + (size_t)y * p.w, src + job.row_stride * y, (size_t)p.w);
                }
            }
        };
        // One thread copies about 10 GB/s, i.e. 13 ms for 64 1080p frames, and nothing runs on the GPU while the
        // first chunk is staged: large chunks are copied by four threads.
        const int helpers = (size_t)m * px >= ((size_t)32 << 20) ? std::min(3, m - 1) : 0;
        if (helpers > 0) {
            std::vector<std::thread> th;
            const int per = (m + helpers) / (helpers + 1);
            for (int t = 1; t <= helpers; t++) th.emplace_back(stage_frames, std::min(m, t * per), std::min(m, (t + 1) * per));
            stage_frames(0, std::min(m, per));
            for (auto &t : th) t.join();
        } else {
            stage_frames(0, m);
        }
        HIP_TRY(hipMemcpyAsync(b->d_frames, stage, (size_t)m * px, hipMemcpyHostToDevice, p.up));
        HIP_TRY(hipEventRecord(p.ev_up[slot], p.up));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, p.ev_up[slot], 0));
        b->d_ext_frames = nullptr;
        int rc = run_pipeline(b, m, nullptr);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(p.ev_run[slot], ctx->stream));
        if (k >= 1) {
            rc = pipe_download(ctx, p, slot ^ 1, job, firsts[k - 1], sizes[k - 1]);
            if (rc) return rc;
        }
    }
    if (nchunks >= 1) {
        const int k = nchunks - 1;
        int rc = pipe_download(ctx, p, k & 1, job, firsts[k], sizes[k]);
        if (rc) return rc;
    }
    return SIFT_OK;
}

}  // namespace

extern "C" {

int sift_shard_bounds(int64_t n, int parts, int64_t *bounds) {
    if (n < 0 || parts < 1 || !bounds) return fail(SIFT_ERR_INVALID, "bad argument");
    for (int r = 0; r <= parts; r++) bounds[r] = n * r / parts;
    return SIFT_OK;
}

int sift_knn2_exchange_plan(int nq, int parts, int rank, int32_t *slice_len, int32_t *own_lo, int32_t *own_hi) {
    if (nq < 0 || parts < 1 || rank < 0 || rank >= parts || !slice_len || !own_lo || !own_hi)
        return fail(SIFT_ERR_INVALID, "bad argument");
    const int len = (nq + parts - 1) / parts;
    *slice_len = len;
    *own_lo = (int32_t)std::min<int64_t>(nq, (int64_t)rank * len);
    *own_hi = (int32_t)std::min<int64_t>(nq, (int64_t)(rank + 1) * len);
    return SIFT_OK;
}

int sift_group_create(sift_ctx *const *ctxs, int ngpu, sift_group **out) {
    if (!ctxs || !out || ngpu < 1) return fail(SIFT_ERR_INVALID, "bad argument");
    *out = nullptr;
    for (int i = 0; i < ngpu; i++) {
        if (!ctxs[i]) return fail(SIFT_ERR_INVALID, "NULL context");
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return fail(SIFT_ERR_INVALID, "a context appears twice in the group");
    }
    sift_group *g = new sift_group();
    g->ctx.assign(ctxs, ctxs + ngpu);
    g->knn.resize((size_t)ngpu);
    g->pipe.resize((size_t)ngpu);
    *out = g;
    return SIFT_OK;
}

int sift_group_destroy(sift_group *g) {
    if (!g) return SIFT_OK;
    for (size_t r = 0; r < g->ctx.size(); r++) {
        hipSetDevice(g->ctx[r]->device);
        hipStreamSynchronize(g->ctx[r]->stream);
        if (g->knn[r].buf) hipFree(g->knn[r].buf);
        pipe_release(g->ctx[r], g->pipe[r]);
    }
    for (ncclComm_t c : g->comms)
        if (c) g->rccl.CommDestroy(c);
    for (hipEvent_t e : g->sent)
        if (e) hipEventDestroy(e);
    delete g;
    return SIFT_OK;
}

int sift_group_size(const sift_group *g) { return g ? (int)g->ctx.size() : 0; }

int sift_frame_results_free(sift_frame_result *results, int n) {
    if (!results) return SIFT_OK;
    for (int i = 0; i < n; i++) {
        std::free(results[i].keypoints);
        std::free(results[i].descriptors);
        results[i].keypoints = nullptr;
        results[i].descriptors = nullptr;
        results[i].n = 0;
    }
    return SIFT_OK;
}

int sift_extract_batch_u8(sift_group *g, const uint8_t *frames, int n, int w, int h, size_t row_stride, size_t frame_stride,
                          int frames_per_launch, sift_frame_result *results) {
    if (!g || (n > 0 && (!frames || !results))) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (n < 0 || w < 2 || h < 2) return fail(SIFT_ERR_INVALID, "frame count or size out of range");
    if (row_stride < (size_t)w || frame_stride < row_stride * (size_t)(h - 1) + (size_t)w)
        return fail(SIFT_ERR_INVALID, "stride smaller than a row / a frame");
    if (n == 0) return SIFT_OK;
    std::memset(results, 0, (size_t)n * sizeof(sift_frame_result));
    const int ngpu = (int)g->ctx.size();
    std::vector<int64_t> bounds((size_t)ngpu + 1);
    sift_shard_bounds(n, ngpu, bounds.data());
    std::vector<int> rc((size_t)ngpu, SIFT_OK);
    std::vector<std::string> msg((size_t)ngpu);
    auto work = [&](int r) {
        const int lo = (int)bounds[r], hi = (int)bounds[r + 1];
        if (lo == hi) return;
        int chunk = frames_per_launch > 0 ? frames_per_launch : frames_per_launch < 0 ? -frames_per_launch : 16;
        chunk = std::min(chunk, hi - lo);
        int e = pipe_prepare(g->ctx[r], g->pipe[r], w, h, std::max(chunk, g->pipe[r].w == w && g->pipe[r].h == h ? g->pipe[r].chunk : 0));
        if (!e) {
            ExtractJob job{frames, row_stride, frame_stride, lo, hi, results, frames_per_launch <= 0, chunk};
            e = pipe_run(g->ctx[r], g->pipe[r], job);
        }
        if (e) {
            rc[r] = e;
            msg[r] = sift_last_error();  // this thread's message
            hipSetDevice(g->ctx[r]->device);  // nothing of this GPU's pipeline is left in flight
            if (g->pipe[r].up) hipStreamSynchronize(g->pipe[r].up);
            hipStreamSynchronize(g->ctx[r]->stream);
            if (g->pipe[r].down) hipStreamSynchronize(g->pipe[r].down);
        }
    };
    if (ngpu == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < ngpu; r++) th.emplace_back(work, r);
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < ngpu; r++)
        if (rc[r]) {
            sift_frame_results_free(results, n);
            return fail(rc[r], "GPU " + std::to_string(g->ctx[r]->device) + ": " + msg[r]);
        }
    return SIFT_OK;
}

int sift_knn2_sharded(sift_group *g, const uint8_t *q, int nq, const uint8_t *t, int nt, sift_knn2 *out, uint8_t *accept) {
    if (!g || (nq > 0 && (!q || !out)) || (nt > 0 && !t)) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (nq < 0 || nt < 0) return fail(SIFT_ERR_INVALID, "negative count");
    if (nq == 0) return SIFT_OK;
    if (any_byte_above_127(q, (size_t)nq * 128)) return fail(SIFT_ERR_NOT_INTEGRAL, "query descriptor value above 127");
    if (any_byte_above_127(t, (size_t)nt * 128)) return fail(SIFT_ERR_NOT_INTEGRAL, "train descriptor value above 127");
    const int ngpu = (int)g->ctx.size();
    // RCCL wants one communicator rank per GPU: a group with members that share a GPU exchanges by copies instead
    bool by_copies = false;
    for (int i = 0; i < ngpu; i++)
        for (int j = 0; j < i; j++)
            if (g->ctx[j]->device == g->ctx[i]->device) by_copies = true;
    int rc = SIFT_OK;
    if (by_copies) {
        while ((int)g->sent.size() < ngpu) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipSetDevice(g->ctx[g->sent.size()]->device));
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            g->sent.push_back(ev);
        }
    } else if ((rc = load_rccl(g->rccl))) {
        return rc;
    }
    if (!by_copies && g->comms.empty()) {
        std::vector<int> devs;
        for (sift_ctx *c : g->ctx) devs.push_back(c->device);
        g->comms.assign((size_t)ngpu, nullptr);
        const ncclResult_t e = g->rccl.CommInitAll(g->comms.data(), ngpu, devs.data());
        if (e != ncclSuccess) {
            g->comms.clear();
            return fail(SIFT_ERR_HIP, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(e));
        }
    }
    // Whatever fails below, no stream of the group is left with work that still reads the caller's buffers.
    struct Drain {
        sift_group *g;
        bool armed = true;
        ~Drain() {
            if (!armed) return;
            for (sift_ctx *c : g->ctx) {
                hipSetDevice(c->device);
                hipStreamSynchronize(c->stream);
            }
        }
    } drain{g};
    std::vector<int64_t> bounds((size_t)ngpu + 1);
    sift_shard_bounds(nt, ngpu, bounds.data());
    int32_t slice = 0, lo = 0, hi = 0;
    sift_knn2_exchange_plan(nq, ngpu, 0, &slice, &lo, &hi);
    const size_t rec = sizeof(sift_knn2);
    const size_t qb = up256((size_t)nq * 128), partb = up256((size_t)ngpu * slice * rec), outb = up256((size_t)slice * rec);
    struct Ptrs {
        uint8_t *dq, *dt;
        char *part, *recv;
        sift_knn2 *dout;
        uint8_t *dacc;
    };
    std::vector<Ptrs> P((size_t)ngpu);
    // 1. operands up, local top-2 with global train indices.  Copies out of pageable memory return when the data has
    // left the caller's buffer, so one host thread would feed the GPUs one after the other: every GPU has its own.
    auto feed = [&](int r) -> int {
        sift_ctx *c = g->ctx[r];
        HIP_TRY(hipSetDevice(c->device));
        const int64_t nr = bounds[r + 1] - bounds[r];
        const size_t tb = up256((size_t)nr * 128);
        const size_t need = qb + tb + 2 * partb + outb + up256((size_t)slice);
        if (need > g->knn[r].bytes) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            const int grow = ensure_bytes(&g->knn[r].buf, &g->knn[r].bytes, need);
            if (grow) return grow;
        }
        char *base = static_cast<char *>(g->knn[r].buf);
        Ptrs &p = P[r];
        p.dq = (uint8_t *)base;
        p.dt = (uint8_t *)(base + qb);
        p.part = base + qb + tb;
        p.recv = p.part + partb;
        p.dout = (sift_knn2 *)(p.recv + partb);
        p.dacc = (uint8_t *)(p.recv + partb + outb);
        HIP_TRY(hipMemcpyAsync(p.dq, q, (size_t)nq * 128, hipMemcpyHostToDevice, c->stream));
        if (nr > 0) HIP_TRY(hipMemcpyAsync(p.dt, t + (size_t)bounds[r] * 128, (size_t)nr * 128, hipMemcpyHostToDevice, c->stream));
        // records past the last query (the last slice is padded to the common length) carry index -1: the merge skips them
        HIP_TRY(hipMemsetAsync(p.part, 0xff, (size_t)ngpu * slice * rec, c->stream));
        return sift_knn2_device(c, p.dq, nq, p.dt, (int)nr, (int)bounds[r], p.part);
    };
    if (ngpu == 1) {
        if ((rc = feed(0))) return rc;
    } else {
        std::vector<int> frc((size_t)ngpu, SIFT_OK);
        std::vector<std::string> fmsg((size_t)ngpu);
        std::vector<std::thread> th;
        for (int r = 0; r < ngpu; r++)
            th.emplace_back([&, r] {
                frc[r] = feed(r);
                if (frc[r]) fmsg[r] = sift_last_error();  // the message is the failing thread's
            });
        for (auto &x : th) x.join();
        for (int r = 0; r < ngpu; r++)
            if (frc[r]) return fail(frc[r], "GPU " + std::to_string(g->ctx[r]->device) + ": " + fmsg[r]);
    }
    // 2. slice s of every GPU's partial results goes to GPU s, which files it under the sender's index
    if (by_copies) {
        for (int r = 0; r < ngpu; r++) {
            HIP_TRY(hipSetDevice(g->ctx[r]->device));
            for (int s = 0; s < ngpu; s++)
                HIP_TRY(hipMemcpyPeerAsync(P[s].recv + (size_t)r * slice * rec, g->ctx[s]->device, P[r].part + (size_t)s * slice * rec,
                                           g->ctx[r]->device, (size_t)slice * rec, g->ctx[r]->stream));
            HIP_TRY(hipEventRecord(g->sent[r], g->ctx[r]->stream));
        }
        for (int s = 0; s < ngpu; s++) {
            HIP_TRY(hipSetDevice(g->ctx[s]->device));
            for (int r = 0; r < ngpu; r++)
                if (r != s) HIP_TRY(hipStreamWaitEvent(g->ctx[s]->stream, g->sent[r], 0));
        }
    } else {
        ncclResult_t e = g->rccl.GroupStart();
        for (int r = 0; r < ngpu && e == ncclSuccess; r++) {
            for (int s = 0; s < ngpu && e == ncclSuccess; s++) {
                e = g->rccl.Send(P[r].part + (size_t)s * slice * rec, (size_t)slice * rec, ncclChar, s, g->comms[r], g->ctx[r]->stream);
                if (e == ncclSuccess)
                    e = g->rccl.Recv(P[r].recv + (size_t)s * slice * rec, (size_t)slice * rec, ncclChar, s, g->comms[r], g->ctx[r]->stream);
            }
        }
        const ncclResult_t e2 = g->rccl.GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) return fail(SIFT_ERR_HIP, std::string("RCCL exchange: ") + g->rccl.GetErrorString(e));
    }
    // 3. merge on the owner of each slice, results down
    for (int r = 0; r < ngpu; r++) {
        sift_ctx *c = g->ctx[r];
        HIP_TRY(hipSetDevice(c->device));
        if ((rc = sift_knn2_merge_device(c, P[r].recv, ngpu, slice, P[r].dout, P[r].dacc))) return rc;
        sift_knn2_exchange_plan(nq, ngpu, r, &slice, &lo, &hi);
        if (hi > lo) {
            HIP_TRY(hipMemcpyAsync(out + lo, P[r].dout, (size_t)(hi - lo) * rec, hipMemcpyDeviceToHost, c->stream));
            if (accept) HIP_TRY(hipMemcpyAsync(accept + lo, P[r].dacc, (size_t)(hi - lo), hipMemcpyDeviceToHost, c->stream));
        }
    }
    for (int r = 0; r < ngpu; r++) {
        HIP_TRY(hipSetDevice(g->ctx[r]->device));
        HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream));
    }
    drain.armed = false;
    return SIFT_OK;
}

}  // extern "C"
