// C ABI of the MI355X SIFT engine: contexts, frame batches, the exec()
// pipeline as a sequence of HIP launches on one stream, result download, and
// the matcher entry points.  See include/sift_amd.h for the contract.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "abi_internal.h"

using namespace sa;

namespace sa {

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

}  // namespace sa

namespace {

// cvRound: round half to even
int cv_round(double v) { return (int)std::lrint(v); }

// getGaussianKernel(n, sigma, CV_64F) as OpenCV 4.x builds it: taps on doubled
// integer coordinates, t = exp(x*x * (-0.125/sigma^2)), sum = 2*sum(t) + 1,
// every tap multiplied by 1/sum, centre tap = 1/sum.  n = cvRound(8*sigma+1)|1
// (GaussianBlur on a non-8-bit depth; sift.cpp:42, :150).
int gaussian_taps(double sigma, Taps &taps) {
    const int n = cv_round(sigma * 4 * 2 + 1) | 1;
    if (n > MAX_TAPS) return -1;
    const int n2 = (n - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double t[MAX_TAPS];
    double sum = 0.0;
    int x = 1 - n;
    for (int i = 0; i < n2; i++, x += 2) {
        t[i] = std::exp((double)(x * x) * scale2x);
        sum += t[i];
    }
    sum *= 2.0;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < MAX_TAPS; i++) taps.c[i] = 0.0;
    for (int i = 0; i < n2; i++) {
        const double v = t[i] * mul1;
        taps.c[i] = v;
        taps.c[n - 1 - i] = v;
    }
    taps.c[n2] = 1.0 * mul1;
    return n;
}

}  // namespace

// Page-locked staging: copies between pageable user memory and the device go through it, because the runtime
// otherwise pins and unpins the user's pages on every call (milliseconds for a frame or a descriptor block).
// The stream must be idle when it is (re)used.
static int ensure_pinned(sift_ctx *ctx, size_t want) {
    if (ctx->pinned_bytes >= want) return SIFT_OK;
    if (ctx->pinned) {
        hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
    return SIFT_OK;
}

// Staging for a host-to-device copy: waits until the stream has drained (so that nothing still reads the buffer),
// then hands out `bytes` of page-locked memory.
static int stage_in(sift_ctx *ctx, size_t bytes, void **out) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int rc = ensure_pinned(ctx, bytes);
    if (rc) return rc;
    *out = ctx->pinned;
    return SIFT_OK;
}

extern "C" {

const char *sift_last_error(void) { return g_last_error.c_str(); }

int sift_device_count(int *count) {
    if (!count) return fail(SIFT_ERR_INVALID, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return SIFT_OK;
}

int sift_ctx_create(int device, sift_ctx **out) {
    if (!out) return fail(SIFT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(SIFT_ERR_INVALID, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    sift_ctx *c = new sift_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(SIFT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    if (pyramid_init_attributes() != 0) {
        hipStreamDestroy(c->stream);
        delete c;
        return fail(SIFT_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    }
    // address of the matcher's static scratch array on this device (the device is current here)
    e = knn_static_scratch(device, &c->knn_static);
    if (e != hipSuccess) {
        hipStreamDestroy(c->stream);
        delete c;
        return fail(SIFT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    }
    *out = c;
    return SIFT_OK;
}

int sift_ctx_destroy(sift_ctx *ctx) {
    if (!ctx) return SIFT_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->cached) sift_batch_destroy(ctx->cached);
    if (ctx->knn_scratch) hipFree(ctx->knn_scratch);
    if (ctx->knn_io) hipFree(ctx->knn_io);
    if (ctx->h_work) hipFree(ctx->h_work);
    if (ctx->st_work) hipFree(ctx->st_work);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->aux) hipStreamDestroy(ctx->aux);
    if (ctx->aux2) hipStreamDestroy(ctx->aux2);
    for (hipEvent_t e : ctx->oct_ready)
        if (e) hipEventDestroy(e);
    if (ctx->aux_done) hipEventDestroy(ctx->aux_done);
    if (ctx->aux2_done) hipEventDestroy(ctx->aux2_done);
    for (hipEvent_t e : ctx->group_ev)
        if (e) hipEventDestroy(e);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return SIFT_OK;
}

int sift_ctx_set_single_image_capacity(sift_ctx *ctx, int max_keypoints) {
    if (!ctx || max_keypoints < 0) return fail(SIFT_ERR_INVALID, "bad argument");
    ctx->single_cap = max_keypoints;
    if (ctx->cached) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        sift_batch_destroy(ctx->cached);
        ctx->cached = nullptr;
    }
    return SIFT_OK;
}

int sift_ctx_stream(sift_ctx *ctx, void **stream) {
    if (!ctx || !stream) return fail(SIFT_ERR_INVALID, "NULL argument");
    *stream = (void *)ctx->stream;
    return SIFT_OK;
}

int sift_ctx_synchronize(sift_ctx *ctx) {
    if (!ctx) return fail(SIFT_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SIFT_OK;
}

int sift_num_octaves(int w, int h) {
    if (w <= 0 || h <= 0) return 0;
    return (int)std::round(std::log2((double)std::min(w, h)));  // sift.cpp:36
}

int sift_octave_dims(int w, int h, int oct, int *ow, int *oh) {
    if (w <= 0 || h <= 0 || oct < 0 || !ow || !oh) return fail(SIFT_ERR_INVALID, "bad argument");
    int cw = 2 * w, ch = 2 * h;  // sift.cpp:40
    for (int o = 0; o < oct; o++) {
        cw = cv_round(cw * 0.5);  // resize(fx=fy=0.5): dsize = cvRound(dim*0.5), sift.cpp:153
        ch = cv_round(ch * 0.5);
    }
    *ow = cw;
    *oh = ch;
    return SIFT_OK;
}

int sift_batch_destroy(sift_batch *b) {
    if (!b) return SIFT_OK;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    for (int o = 0; o < MAX_OCTAVES; o++) {
        if (b->pyr.gauss[o]) hipFree(b->pyr.gauss[o]);
        if (b->pyr.grad[o]) hipFree(b->pyr.grad[o]);
    }
    hipFree(b->d_frames);
    hipFree(b->d_bgr);
    hipFree(b->d_cand);
    hipFree(b->d_oriented);
    hipFree(b->d_sorted);
    hipFree(b->d_final);
    hipFree(b->d_flags);
    hipFree(b->d_desc);
    hipFree(b->d_counters);
    hipFree(b->d_cursors);
    hipFree(b->d_work);
    for (hipEvent_t e : b->events) hipEventDestroy(e);
    if (b->ctx->cached == b) b->ctx->cached = nullptr;
    delete b;
    return SIFT_OK;
}

int sift_batch_create(sift_ctx *ctx, int w, int h, int max_frames, int max_keypoints_per_frame, sift_batch **out) {
    if (!ctx || !out) return fail(SIFT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (w < 2 || h < 2 || max_frames < 1) return fail(SIFT_ERR_INVALID, "frame size or count out of range");
    if ((int64_t)w * h > (int64_t)1 << 28) return fail(SIFT_ERR_INVALID, "frame too large");
    const int octaves = sift_num_octaves(w, h);
    if (octaves < 1 || octaves > MAX_OCTAVES) return fail(SIFT_ERR_INVALID, "octave count out of range");
    HIP_TRY(hipSetDevice(ctx->device));
    sift_batch *b = new sift_batch();
    b->ctx = ctx;
    b->w = w;
    b->h = h;
    b->max_frames = max_frames;
    b->pyr.octaves = octaves;
    // kernel taps: base blur (sift.cpp:41) and the incremental sigmas (sift.cpp:134-141)
    {
        const double diff = std::max(std::sqrt(SIGMA * SIGMA - 4 * (ASSUMED_BLUR * ASSUMED_BLUR)), 0.1);
        b->base_ntaps = gaussian_taps(diff, b->base_taps);
        const double k = std::pow(2, 1.0 / SCALES);
        double prev = SIGMA;
        for (int i = 1; i < IMAGES; i++) {
            const double now = prev * k;
            const double sig = std::sqrt(now * now - prev * prev);
            b->lvl_ntaps[i] = gaussian_taps(sig, b->lvl_taps[i]);
            prev = now;
        }
        bool ok = b->base_ntaps == 11;
        for (int i = 1; i < IMAGES; i++) ok = ok && b->lvl_ntaps[i] > 0 && blur_radius_supported(b->lvl_ntaps[i] / 2);
        if (!ok) {
            delete b;
            return fail(SIFT_ERR_INVALID, "unexpected Gaussian tap count");
        }
    }
    const int64_t px = (int64_t)w * h;
    int cap = max_keypoints_per_frame > 0 ? max_keypoints_per_frame : (int)std::max<int64_t>(8192, px / 16);
    b->cand_cap = cap;
    b->kp_cap = cap;
    auto cleanup = [&](hipError_t e, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        sift_batch_destroy(b);
        return fail(SIFT_ERR_HIP, msg);
    };
    hipError_t e;
    for (int o = 0; o < octaves; o++) {
        int ow, oh;
        sift_octave_dims(w, h, o, &ow, &oh);
        if (ow < 1 || oh < 1) {
            sift_batch_destroy(b);
            return fail(SIFT_ERR_INVALID, "pyramid octave collapsed to zero size");
        }
        b->pyr.w[o] = ow;
        b->pyr.h[o] = oh;
        const size_t plane = (size_t)ow * oh * sizeof(double);
        if ((e = hipMalloc((void **)&b->pyr.gauss[o], plane * IMAGES * max_frames)) != hipSuccess)
            return cleanup(e, "hipMalloc(gauss)");
        // gradient planes only where the scan range is non-empty (sift.cpp:188, :233)
        if (ow > 2 * BORDER && oh > 2 * BORDER &&
            (e = hipMalloc((void **)&b->pyr.grad[o], (size_t)ow * oh * sizeof(Grad) * SCALES * max_frames)) != hipSuccess)
            return cleanup(e, "hipMalloc(grad)");
    }
    const size_t F = (size_t)max_frames;
    if ((e = hipMalloc((void **)&b->d_frames, F * px)) != hipSuccess) return cleanup(e, "hipMalloc(frames)");
    if ((e = hipMalloc((void **)&b->d_cand, F * cap * sizeof(Candidate))) != hipSuccess) return cleanup(e, "hipMalloc(cand)");
    if ((e = hipMalloc((void **)&b->d_oriented, F * cap * sizeof(DevKeypoint))) != hipSuccess) return cleanup(e, "hipMalloc(kp)");
    if ((e = hipMalloc((void **)&b->d_sorted, F * cap * sizeof(DevKeypoint))) != hipSuccess) return cleanup(e, "hipMalloc(kp)");
    if ((e = hipMalloc((void **)&b->d_final, F * cap * sizeof(DevKeypoint))) != hipSuccess) return cleanup(e, "hipMalloc(kp)");
    if ((e = hipMalloc((void **)&b->d_flags, F * cap * 2)) != hipSuccess) return cleanup(e, "hipMalloc(flags)");
    if ((e = hipMalloc((void **)&b->d_desc, F * cap * 128)) != hipSuccess) return cleanup(e, "hipMalloc(desc)");
    if ((e = hipMalloc((void **)&b->d_counters, F * sizeof(FrameCounters))) != hipSuccess) return cleanup(e, "hipMalloc(counters)");
    if ((e = hipMalloc((void **)&b->d_cursors, 3 * F * sizeof(int32_t))) != hipSuccess) return cleanup(e, "hipMalloc(cursors)");
    if ((e = hipMemset(b->d_counters, 0, F * sizeof(FrameCounters))) != hipSuccess) return cleanup(e, "hipMemset");
    *out = b;
    return SIFT_OK;
}

int sift_batch_frames_device_ptr(sift_batch *b, void **dptr) {
    if (!b || !dptr) return fail(SIFT_ERR_INVALID, "NULL argument");
    *dptr = b->d_frames;
    return SIFT_OK;
}

int sift_batch_set_external_frames(sift_batch *b, const void *d_frames) {
    if (!b) return fail(SIFT_ERR_INVALID, "batch is NULL");
    b->d_ext_frames = (const uint8_t *)d_frames;
    return SIFT_OK;
}

int sift_batch_upload_gray8(sift_batch *b, int frame, const uint8_t *gray, size_t stride) {
    if (!b || !gray) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (frame < 0 || frame >= b->max_frames) return fail(SIFT_ERR_INVALID, "frame index out of range");
    if (stride < (size_t)b->w) return fail(SIFT_ERR_INVALID, "stride smaller than width");
    HIP_TRY(hipSetDevice(b->ctx->device));
    uint8_t *dst = b->d_frames + (size_t)frame * b->w * b->h;
    void *stage = nullptr;
    int rc = stage_in(b->ctx, (size_t)b->w * b->h, &stage);
    if (rc) return rc;
    for (int y = 0; y < b->h; y++) std::memcpy(static_cast<char *>(stage) + (size_t)y * b->w, gray + stride * y, (size_t)b->w);
    HIP_TRY(hipMemcpyAsync(dst, stage, (size_t)b->w * b->h, hipMemcpyHostToDevice, b->ctx->stream));
    return SIFT_OK;
}

int sift_batch_upload_bgr8(sift_batch *b, int frame, const uint8_t *bgr, size_t stride) {
    if (!b || !bgr) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (frame < 0 || frame >= b->max_frames) return fail(SIFT_ERR_INVALID, "frame index out of range");
    if (stride < (size_t)b->w * 3) return fail(SIFT_ERR_INVALID, "stride smaller than 3*width");
    HIP_TRY(hipSetDevice(b->ctx->device));
    const size_t px = (size_t)b->w * b->h;
    if (!b->d_bgr) HIP_TRY(hipMalloc((void **)&b->d_bgr, px * 3));
    void *stage = nullptr;
    int rc = stage_in(b->ctx, px * 3, &stage);  // waits for the stream: the device-side BGR buffer is free again too
    if (rc) return rc;
    for (int y = 0; y < b->h; y++) std::memcpy(static_cast<char *>(stage) + (size_t)y * b->w * 3, bgr + stride * y, (size_t)b->w * 3);
    HIP_TRY(hipMemcpyAsync(b->d_bgr, stage, px * 3, hipMemcpyHostToDevice, b->ctx->stream));
    launch_bgr_to_gray(b->ctx->stream, b->d_bgr, b->d_frames + (size_t)frame * px, px);
    HIP_TRY(hipGetLastError());
    return SIFT_OK;
}

}  // extern "C"

namespace {

struct StageTimer {
    sift_batch *b;
    bool on;
    size_t used = 0;
    std::vector<int> stage_of;
    hipStream_t st;
    int begin(int stage) {
        if (!on) return 0;
        while (b->events.size() < used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            b->events.push_back(e);
        }
        stage_of.push_back(stage);
        return hipEventRecord(b->events[used], st) == hipSuccess ? 0 : -1;
    }
    int end() {
        if (!on) return 0;
        int rc = hipEventRecord(b->events[used + 1], st) == hipSuccess ? 0 : -1;
        used += 2;
        return rc;
    }
};

}  // namespace

// Side streams and their events, created together: the context keeps them only if every creation succeeded.
static int ensure_aux(sift_ctx *ctx) {
    if (ctx->aux) return SIFT_OK;
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ready[MAX_OCTAVES] = {};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking);
    for (int i = 0; i < MAX_OCTAVES && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ready[i], hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        for (hipEvent_t ev : ready)
            if (ev) hipEventDestroy(ev);
        for (int i = 0; i < 2; i++) {
            if (done[i]) hipEventDestroy(done[i]);
            if (aux[i]) hipStreamDestroy(aux[i]);
        }
        return fail(SIFT_ERR_HIP, std::string("side stream set-up: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < MAX_OCTAVES; i++) ctx->oct_ready[i] = ready[i];
    ctx->aux_done = done[0];
    ctx->aux2_done = done[1];
    ctx->aux2 = aux[1];
    ctx->aux = aux[0];
    return SIFT_OK;
}

int sa::pipeline_group_times(sift_ctx *ctx, double ms[5]) {
    float t[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventElapsedTime(&t[i], ctx->group_ev[i], ctx->group_ev[i + 1]));
    ms[0] = t[0];  // gen_gaussian_images (base blur included: the constructor's share is not separable from the launch order)
    ms[1] = 0.0;   // gen_dog_images: no such pass, the differences are formed in registers by their readers
    ms[2] = t[1];  // gen_scale_space_extrema
    ms[3] = t[2];  // clean_keypoints
    ms[4] = t[3];  // get_descriptors
    return SIFT_OK;
}

int sa::run_pipeline(sift_batch *b, int n, sift_stage_times *times, bool group_events) {
    sift_ctx *ctx = b->ctx;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->pipeline_runs++;
    StageTimer tm{b, times != nullptr, 0, {}, st};
    HIP_TRY(hipMemsetAsync(b->d_counters, 0, (size_t)n * sizeof(FrameCounters), st));
    HIP_TRY(hipMemsetAsync(b->d_cursors, 0, 3 * (size_t)b->max_frames * sizeof(int32_t), st));
#define TIMED(stage, call)                                              \
    do {                                                                \
        if (tm.begin(stage)) return fail(SIFT_ERR_HIP, "hipEventRecord"); \
        call;                                                           \
        if (tm.end()) return fail(SIFT_ERR_HIP, "hipEventRecord");      \
    } while (0)
    const uint8_t *frames = b->d_ext_frames ? b->d_ext_frames : b->d_frames;
    // A workload that does not fill the chip (a single frame, a few small ones) is bound by the latency of its
    // ~70 dependent launches.  Only levels 1-3 of an octave feed the next octave, so the main stream runs base and
    // levels 1-3 octave after octave while two side streams take levels 4-5, the gradient planes and the extrema scan
    // of each finished octave: about half as many launches on the critical path.  The per-keypoint stages run the
    // variants that give a keypoint more lanes.  Large batches keep the plain order (their launches fill the chip,
    // and the profiled pass times each launch on its own).
    const bool two_streams = !times && (int64_t)n * b->w * b->h <= ((int64_t)1 << 22);
    group_events = group_events && !times;
    if (group_events) {
        for (hipEvent_t &e : ctx->group_ev)
            if (!e) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(ctx->group_ev[0], st));
    }
    if (two_streams) {
        int rc = ensure_aux(ctx);
        if (rc) return rc;
        hipStream_t side[2] = {ctx->aux, ctx->aux2};
        // Once a side stream has work, every way out of this block joins both first: later calls on the main
        // stream must not overlap kernels that are still writing the pyramid and the candidate lists.
        hipError_t e = hipSuccess;
        launch_base(st, frames, b->w, b->h, n, b->pyr, b->base_taps, b->base_ntaps);
        // The upper octaves are a few thousand pixels each: all their levels are one launch on the main stream
        // (behind level 3 of the last large octave, which writes their first level 0), followed by their scans.
        // The main stream's launches are all enqueued before the second stream's: the host needs longer to
        // enqueue a small launch than the GPU to run it, and the chain that everything waits for is the main one.
        int o_small = first_small_octave(b->pyr), o_large = b->pyr.octaves;
        for (int o = 0; o < b->pyr.octaves && e == hipSuccess; o++) {
            if (o == o_small) {
                if (launch_small_octaves(st, b->pyr, o_small, n, b->lvl_taps, b->lvl_ntaps)) {
                    launch_extrema_gradients_from(st, b->pyr, o_small, n, b->d_cand, b->cand_cap, b->d_counters);
                    o_large = o_small;
                    break;
                }
                o_small = b->pyr.octaves;  // not taken: these octaves go level by level like the others
            }
            for (int lvl = 1; lvl <= IMAGES - 3; lvl++) launch_blur(st, b->pyr, o, lvl, n, b->lvl_taps[lvl], b->lvl_ntaps[lvl]);
            e = hipEventRecord(ctx->oct_ready[o], st);
        }
        // levels 4-5 and the scan of an octave depend on nothing of the other octaves: even and odd octaves take turns
        // on the two side streams
        for (int o = 0; o < o_large && e == hipSuccess; o++) {
            hipStream_t aux = side[o & 1];
            if ((e = hipStreamWaitEvent(aux, ctx->oct_ready[o], 0)) != hipSuccess) break;
            for (int lvl = IMAGES - 2; lvl < IMAGES; lvl++) launch_blur(aux, b->pyr, o, lvl, n, b->lvl_taps[lvl], b->lvl_ntaps[lvl]);
            launch_extrema_gradients(aux, b->pyr, o, n, b->d_cand, b->cand_cap, b->d_counters);
        }
        if (e == hipSuccess && group_events) e = hipEventRecord(ctx->group_ev[1], st);
        if (e == hipSuccess) e = hipEventRecord(ctx->aux_done, side[0]);
        if (e == hipSuccess) e = hipEventRecord(ctx->aux2_done, side[1]);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->aux_done, 0);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->aux2_done, 0);
        if (e != hipSuccess) {
            hipStreamSynchronize(side[0]);
            hipStreamSynchronize(side[1]);
            return fail(SIFT_ERR_HIP, std::string("side-stream schedule: ") + hipGetErrorString(e));
        }
    } else {
        TIMED(SIFT_STAGE_BASE, launch_base(st, frames, b->w, b->h, n, b->pyr, b->base_taps, b->base_ntaps));
        for (int o = 0; o < b->pyr.octaves; o++) {
            int lvl = 1;
            // Levels 1 and 2 of a large octave are one launch that never reads level 1 back (0.33 ms of 4.68 per 64
            // 1080p frames; the single-frame schedule keeps one launch per level: it is bound by launch latency).
            if (blur_levels_1_2_pair(b->pyr, o, b->lvl_ntaps)) {
                TIMED(SIFT_STAGE_BLUR_L12, launch_blur_levels_1_2(st, b->pyr, o, n, b->lvl_taps));
                lvl = 3;
            }
            for (; lvl < IMAGES; lvl++)
                TIMED(SIFT_STAGE_BLUR_L1 + lvl - 1, launch_blur(st, b->pyr, o, lvl, n, b->lvl_taps[lvl], b->lvl_ntaps[lvl]));
        }
        if (group_events) HIP_TRY(hipEventRecord(ctx->group_ev[1], st));
        for (int o = 0; o < b->pyr.octaves; o++)
            if (b->pyr.grad[o])
                TIMED(SIFT_STAGE_EXTREMA, launch_extrema_gradients(st, b->pyr, o, n, b->d_cand, b->cand_cap, b->d_counters));
    }
    TIMED(SIFT_STAGE_LOCALIZE, launch_localize_orient(st, b->pyr, n, b->d_cand, b->cand_cap, b->d_sorted, b->d_oriented,
                                                      b->kp_cap, b->d_counters, b->d_cursors + 2 * b->max_frames, two_streams));
    if (group_events) HIP_TRY(hipEventRecord(ctx->group_ev[2], st));
    TIMED(SIFT_STAGE_CLEAN,
          launch_clean(st, n, b->d_oriented, b->d_sorted, b->d_final, b->d_flags, b->kp_cap, b->d_counters,
                       // the descriptor array is not written before the next stage: it takes the partial ranks
                       two_streams ? reinterpret_cast<int32_t *>(b->d_desc) : nullptr));
    if (group_events) HIP_TRY(hipEventRecord(ctx->group_ev[3], st));
    TIMED(SIFT_STAGE_DESCRIPTOR, launch_descriptors(st, b->pyr, n, b->d_final, b->kp_cap, b->d_counters,
                                                    b->d_cursors + b->max_frames, b->d_desc,
                                                    times ? ((int64_t)1 << 40) : (int64_t)n * b->w * b->h));
    if (group_events) HIP_TRY(hipEventRecord(ctx->group_ev[4], st));
#undef TIMED
    HIP_TRY(hipGetLastError());
    if (times) {
        HIP_TRY(hipStreamSynchronize(st));
        std::memset(times, 0, sizeof(*times));
        for (size_t i = 0; i < tm.stage_of.size(); i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, b->events[2 * i], b->events[2 * i + 1]));
            times->ms[tm.stage_of[i]] += ms;
            times->launches[tm.stage_of[i]] += 1;
        }
    }
    return SIFT_OK;
}

extern "C" {

int sift_ctx_pipeline_runs(sift_ctx *ctx, int64_t *runs) {
    if (!ctx || !runs) return fail(SIFT_ERR_INVALID, "NULL argument");
    *runs = ctx->pipeline_runs;
    return SIFT_OK;
}

int sift_batch_run(sift_batch *b, int n_frames) {
    if (!b) return fail(SIFT_ERR_INVALID, "batch is NULL");
    if (n_frames < 1 || n_frames > b->max_frames) return fail(SIFT_ERR_INVALID, "n_frames out of range");
    return run_pipeline(b, n_frames, nullptr);
}

int sift_batch_run_profiled(sift_batch *b, int n_frames, sift_stage_times *times) {
    if (!b || !times) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (n_frames < 1 || n_frames > b->max_frames) return fail(SIFT_ERR_INVALID, "n_frames out of range");
    return run_pipeline(b, n_frames, times);
}

int sift_batch_counts(sift_batch *b, int n_frames, sift_frame_counts *counts) {
    if (!b || !counts) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (n_frames < 1 || n_frames > b->max_frames) return fail(SIFT_ERR_INVALID, "n_frames out of range");
    HIP_TRY(hipSetDevice(b->ctx->device));
    static_assert(sizeof(sift_frame_counts) == sizeof(FrameCounters), "layout");
    HIP_TRY(hipMemcpyAsync(counts, b->d_counters, (size_t)n_frames * sizeof(FrameCounters), hipMemcpyDeviceToHost,
                           b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    bool overflow = false;
    for (int f = 0; f < n_frames; f++) {
        if (counts[f].candidates > b->cand_cap || counts[f].oriented > b->kp_cap) counts[f].overflow = 1;
        overflow = overflow || counts[f].overflow;
    }
    if (overflow) return fail(SIFT_ERR_OVERFLOW, "a frame produced more candidates or keypoints than the batch capacity");
    return SIFT_OK;
}

int sift_batch_download(sift_batch *b, int frame, int capacity, sift_keypoint *keypoints, uint8_t *desc_u8,
                        double *desc_f64, int32_t *n_out) {
    if (!b) return fail(SIFT_ERR_INVALID, "batch is NULL");
    if (frame < 0 || frame >= b->max_frames) return fail(SIFT_ERR_INVALID, "frame index out of range");
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    FrameCounters fc;
    HIP_TRY(hipMemcpyAsync(&fc, b->d_counters + frame, sizeof(fc), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int n = fc.keypoints;
    if (n_out) *n_out = n;
    if (n > capacity) return fail(SIFT_ERR_INVALID, "capacity smaller than the frame's keypoint count");
    if (n == 0) return SIFT_OK;
    // one staged transfer: keypoints, then descriptors, into page-locked memory; converted from there
    const size_t kp_bytes = keypoints ? (size_t)n * sizeof(DevKeypoint) : 0;
    const size_t kp_span = (kp_bytes + 255) & ~(size_t)255;
    const size_t desc_bytes = (desc_u8 || desc_f64) ? (size_t)n * 128 : 0;
    int rc = ensure_pinned(b->ctx, kp_span + desc_bytes);
    if (rc) return rc;
    char *stage = static_cast<char *>(b->ctx->pinned);
    if (kp_bytes)
        HIP_TRY(hipMemcpyAsync(stage, b->d_final + (size_t)frame * b->kp_cap, kp_bytes, hipMemcpyDeviceToHost, st));
    if (desc_bytes)
        HIP_TRY(hipMemcpyAsync(stage + kp_span, b->d_desc + (size_t)frame * b->kp_cap * 128, desc_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (keypoints) {
        static_assert(sizeof(DevKeypoint) >= sizeof(sift_keypoint), "device record holds the public fields first");
        const DevKeypoint *src = reinterpret_cast<const DevKeypoint *>(stage);
        for (int i = 0; i < n; i++) std::memcpy(&keypoints[i], &src[i], sizeof(sift_keypoint));
    }
    if (desc_bytes) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(stage + kp_span);
        if (desc_u8) std::memcpy(desc_u8, src, desc_bytes);
        if (desc_f64)
            for (size_t i = 0; i < desc_bytes; i++) desc_f64[i] = (double)src[i];
    }
    return SIFT_OK;
}

int sift_batch_descriptor_work(sift_batch *b, int n_frames, int64_t *work) {
    if (!b || !work) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (n_frames < 1 || n_frames > b->max_frames) return fail(SIFT_ERR_INVALID, "n_frames out of range");
    sift_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n_frames * 2 * sizeof(unsigned long long);
    if (!b->d_work) HIP_TRY(hipMalloc((void **)&b->d_work, (size_t)b->max_frames * 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(b->d_work, 0, bytes, ctx->stream));
    HIP_TRY(hipMemsetAsync(b->d_cursors + b->max_frames, 0, (size_t)b->max_frames * sizeof(int32_t), ctx->stream));
    launch_descriptor_work(ctx->stream, b->pyr, n_frames, b->d_final, b->kp_cap, b->d_counters, b->d_cursors + b->max_frames,
                           b->d_work);
    HIP_TRY(hipGetLastError());
    static_assert(sizeof(int64_t) == sizeof(unsigned long long), "layout");
    HIP_TRY(hipMemcpyAsync(work, b->d_work, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SIFT_OK;
}

int sift_batch_result_device_ptrs(sift_batch *b, int frame, void **keypoints, void **desc_u8) {
    if (!b) return fail(SIFT_ERR_INVALID, "batch is NULL");
    if (frame < 0 || frame >= b->max_frames) return fail(SIFT_ERR_INVALID, "frame index out of range");
    if (keypoints) *keypoints = b->d_final + (size_t)frame * b->kp_cap;
    if (desc_u8) *desc_u8 = b->d_desc + (size_t)frame * b->kp_cap * 128;
    return SIFT_OK;
}

int sift_batch_get_level(sift_batch *b, int frame, int oct, int lvl, int kind, double *out) {
    if (!b || !out) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (frame < 0 || frame >= b->max_frames || oct < 0 || oct >= b->pyr.octaves)
        return fail(SIFT_ERR_INVALID, "frame or octave out of range");
    if (kind == SIFT_LEVEL_GRAD_MAG || kind == SIFT_LEVEL_GRAD_DIR) {
        if (lvl < 1 || lvl > SCALES) return fail(SIFT_ERR_INVALID, "gradient planes exist for levels 1..3");
        if (!b->pyr.grad[oct]) return fail(SIFT_ERR_INVALID, "this octave is too small to be scanned: it has no gradient planes");
        HIP_TRY(hipSetDevice(b->ctx->device));
        const size_t px = (size_t)b->pyr.w[oct] * b->pyr.h[oct];
        void *stage = nullptr;
        int rc = stage_in(b->ctx, px * sizeof(Grad), &stage);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(stage, b->pyr.grad[oct] + grad_off(b->pyr, oct, frame, lvl), px * sizeof(Grad), hipMemcpyDeviceToHost,
                               b->ctx->stream));
        HIP_TRY(hipStreamSynchronize(b->ctx->stream));
        const Grad *g = static_cast<const Grad *>(stage);
        for (size_t i = 0; i < px; i++) out[i] = kind == SIFT_LEVEL_GRAD_MAG ? g[i].mag : g[i].ori;
        return SIFT_OK;
    }
    const int nl = kind == SIFT_LEVEL_GAUSS ? IMAGES : NDOG;
    if ((kind != SIFT_LEVEL_GAUSS && kind != SIFT_LEVEL_DOG) || lvl < 0 || lvl >= nl)
        return fail(SIFT_ERR_INVALID, "level out of range");
    HIP_TRY(hipSetDevice(b->ctx->device));
    const size_t plane = (size_t)b->pyr.w[oct] * b->pyr.h[oct];
    // images[oct][lvl] = gauss_images[oct][lvl + 1] - gauss_images[oct][lvl] (sift.cpp:173) is not kept on the device:
    // both Gaussian levels come down and are subtracted here, the same IEEE operation the kernels perform
    const int ncopy = kind == SIFT_LEVEL_DOG ? 2 : 1;
    const double *src = b->pyr.gauss[oct] + gauss_off(b->pyr, oct, frame, lvl);  // levels are adjacent planes
    void *stage = nullptr;
    int rc = stage_in(b->ctx, ncopy * plane * sizeof(double), &stage);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stage, src, ncopy * plane * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    const double *lo = static_cast<const double *>(stage), *hi = lo + plane;
    if (kind == SIFT_LEVEL_DOG) {
        for (size_t i = 0; i < plane; i++) out[i] = hi[i] - lo[i];
    } else {
        std::memcpy(out, stage, plane * sizeof(double));
    }
    return SIFT_OK;
}

int sift_batch_get_candidates(sift_batch *b, int frame, int capacity, int32_t *quads, int32_t *n_out) {
    if (!b || !n_out) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (frame < 0 || frame >= b->max_frames) return fail(SIFT_ERR_INVALID, "frame index out of range");
    HIP_TRY(hipSetDevice(b->ctx->device));
    FrameCounters fc;
    HIP_TRY(hipMemcpyAsync(&fc, b->d_counters + frame, sizeof(fc), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    const int n = std::min(fc.candidates, b->cand_cap);
    *n_out = n;
    if (!quads || n == 0) return SIFT_OK;
    if (n > capacity) return fail(SIFT_ERR_INVALID, "capacity too small");
    static_assert(sizeof(Candidate) == 16, "layout");
    HIP_TRY(hipMemcpyAsync(quads, b->d_cand + (size_t)frame * b->cand_cap, (size_t)n * sizeof(Candidate),
                           hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return SIFT_OK;
}

int sift_batch_get_oriented(sift_batch *b, int frame, int capacity, sift_keypoint *kps, int32_t *n_out) {
    if (!b || !n_out) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (frame < 0 || frame >= b->max_frames) return fail(SIFT_ERR_INVALID, "frame index out of range");
    HIP_TRY(hipSetDevice(b->ctx->device));
    FrameCounters fc;
    HIP_TRY(hipMemcpyAsync(&fc, b->d_counters + frame, sizeof(fc), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    const int n = std::min(fc.oriented, b->kp_cap);
    *n_out = n;
    if (!kps || n == 0) return SIFT_OK;
    if (n > capacity) return fail(SIFT_ERR_INVALID, "capacity too small");
    std::vector<DevKeypoint> tmp((size_t)n);
    HIP_TRY(hipMemcpyAsync(tmp.data(), b->d_oriented + (size_t)frame * b->kp_cap, (size_t)n * sizeof(DevKeypoint),
                           hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    for (int i = 0; i < n; i++) std::memcpy(&kps[i], &tmp[i], sizeof(sift_keypoint));
    return SIFT_OK;
}

// ---- single image ---------------------------------------------------------

// The single-image workspace of a context for w x h frames, created or grown on demand.
static int single_workspace(sift_ctx *ctx, int w, int h, int want_cap, bool must_grow, sift_batch **out) {
    sift_batch *b = ctx->cached;
    if (!b || b->w != w || b->h != h || (must_grow && b->kp_cap < want_cap)) {
        if (b) sift_batch_destroy(b);
        ctx->cached = nullptr;
        int rc = sift_batch_create(ctx, w, h, 1, want_cap, &b);
        if (rc) return rc;
        ctx->cached = b;
    }
    *out = b;
    return SIFT_OK;
}

// One host frame into slot 0 of a workspace through the page-locked staging buffer.
static int upload_single(sift_ctx *ctx, sift_batch *b, const uint8_t *img, size_t stride, bool bgr) {
    const int w = b->w, h = b->h;
    const size_t row_bytes = (size_t)w * (bgr ? 3 : 1);
    if (stride < row_bytes) return fail(SIFT_ERR_INVALID, "stride smaller than a row");
    HIP_TRY(hipSetDevice(ctx->device));
    void *stage = nullptr;
    int rc = stage_in(ctx, row_bytes * h, &stage);
    if (rc) return rc;
    for (int y = 0; y < h; y++) std::memcpy(static_cast<char *>(stage) + row_bytes * y, img + stride * y, row_bytes);
    if (bgr) {
        if (!b->d_bgr) HIP_TRY(hipMalloc((void **)&b->d_bgr, row_bytes * h));
        HIP_TRY(hipMemcpyAsync(b->d_bgr, ctx->pinned, row_bytes * h, hipMemcpyHostToDevice, ctx->stream));
        launch_bgr_to_gray(ctx->stream, b->d_bgr, b->d_frames, (size_t)w * h);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(b->d_frames, ctx->pinned, row_bytes * h, hipMemcpyHostToDevice, ctx->stream));
    }
    return SIFT_OK;
}

static int extract_common(sift_ctx *ctx, const uint8_t *img, int w, int h, size_t stride, bool bgr, sift_result *out,
                          double *stage_ms = nullptr) {
    if (!ctx || !img || !out) return fail(SIFT_ERR_INVALID, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    // The reference's lists grow without bound; here a frame whose candidates or keypoints exceed the
    // workspace is run again in a larger one (the counters keep counting past the capacity).
    sift_batch *b = nullptr;
    sift_frame_counts fc;
    int want_cap = ctx->single_cap;
    for (int attempt = 0;; attempt++) {
        int rc = single_workspace(ctx, w, h, want_cap, attempt > 0, &b);
        if (rc) return rc;
        if ((rc = upload_single(ctx, b, img, stride, bgr))) return rc;
        b->d_ext_frames = nullptr;
        if ((rc = run_pipeline(b, 1, nullptr, stage_ms != nullptr))) return rc;
        rc = sift_batch_counts(b, 1, &fc);
        if (rc == SIFT_ERR_OVERFLOW && attempt < 4) {
            const int64_t seen = std::max<int64_t>(std::max(fc.candidates, fc.oriented), b->kp_cap);
            if (seen > (int64_t)1 << 28) return rc;
            want_cap = (int)(2 * seen);
            continue;
        }
        if (rc) return rc;
        break;
    }
    if (stage_ms) {
        int rc = pipeline_group_times(ctx, stage_ms);  // the stream was synchronised by sift_batch_counts
        if (rc) return rc;
    }
    out->counts = fc;
    out->octaves = b->pyr.octaves;
    out->n = fc.keypoints;
    if (fc.keypoints > 0) {
        out->keypoints = (sift_keypoint *)std::malloc(sizeof(sift_keypoint) * (size_t)fc.keypoints);
        out->descriptors = (double *)std::malloc(sizeof(double) * 128 * (size_t)fc.keypoints);
        if (!out->keypoints || !out->descriptors) {
            sift_result_free(out);
            return fail(SIFT_ERR_INVALID, "out of host memory");
        }
        int32_t n = 0;
        const int rc = sift_batch_download(b, 0, fc.keypoints, out->keypoints, nullptr, out->descriptors, &n);
        if (rc) {
            sift_result_free(out);
            return rc;
        }
    }
    return SIFT_OK;
}

int sift_extract_gray8(sift_ctx *ctx, const uint8_t *gray, int w, int h, size_t stride, sift_result *out) {
    return extract_common(ctx, gray, w, h, stride, false, out);
}

int sift_extract_bgr8(sift_ctx *ctx, const uint8_t *bgr, int w, int h, size_t stride, sift_result *out) {
    return extract_common(ctx, bgr, w, h, stride, true, out);
}

int sift_extract_gray8_stages(sift_ctx *ctx, const uint8_t *gray, int w, int h, size_t stride, sift_result *out,
                              double stage_ms[5]) {
    if (!stage_ms) return fail(SIFT_ERR_INVALID, "stage_ms is NULL");
    return extract_common(ctx, gray, w, h, stride, false, out, stage_ms);
}

int sift_base_image_gray8(sift_ctx *ctx, const uint8_t *gray, int w, int h, size_t stride, double *base) {
    if (!ctx || !gray || !base) return fail(SIFT_ERR_INVALID, "NULL argument");
    sift_batch *b = nullptr;
    int rc = single_workspace(ctx, w, h, ctx->single_cap, false, &b);
    if (rc) return rc;
    if ((rc = upload_single(ctx, b, gray, stride, false))) return rc;
    launch_base(ctx->stream, b->d_frames, b->w, b->h, 1, b->pyr, b->base_taps, b->base_ntaps);
    HIP_TRY(hipGetLastError());
    return sift_batch_get_level(b, 0, 0, 0, SIFT_LEVEL_GAUSS, base);
}

int sift_result_free(sift_result *r) {
    if (!r) return SIFT_OK;
    std::free(r->keypoints);
    std::free(r->descriptors);
    r->keypoints = nullptr;
    r->descriptors = nullptr;
    r->n = 0;
    return SIFT_OK;
}

int sift_result_get_f32(const sift_result *r, float *out) {
    if (!r || (!out && r->n > 0)) return fail(SIFT_ERR_INVALID, "NULL argument");
    for (size_t i = 0; i < (size_t)r->n * 128; i++) out[i] = (float)r->descriptors[i];  // sift.cpp:108
    return SIFT_OK;
}

// ---- matcher ----------------------------------------------------------------

}  // extern "C"

int sa::ensure_bytes(void **buf, size_t *have, size_t want) {
    if (*have >= want) return SIFT_OK;
    if (*buf) {
        hipError_t e = hipFree(*buf);
        *buf = nullptr;
        *have = 0;
        if (e != hipSuccess) return fail(SIFT_ERR_HIP, std::string("hipFree: ") + hipGetErrorString(e));
    }
    HIP_TRY(hipMalloc(buf, want));
    *have = want;
    return SIFT_OK;
}

extern "C" {

int sift_knn2_device(sift_ctx *ctx, const void *d_q, int nq, const void *d_t, int nt, int train_index_base,
                     void *d_out) {
    if (!ctx || (nq > 0 && (!d_q || !d_out)) || (nt > 0 && !d_t)) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (nq < 0 || nt < 0) return fail(SIFT_ERR_INVALID, "negative count");
    if (nq == 0) return SIFT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t need = knn_scratch_bytes(nq, nt);
    if (need > ctx->knn_scratch_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        int rc = ensure_bytes(&ctx->knn_scratch, &ctx->knn_scratch_bytes, need);
        if (rc) return rc;
    }
    launch_knn2(ctx->stream, (const uint8_t *)d_q, nq, (const uint8_t *)d_t, nt, train_index_base, (sift_knn2 *)d_out,
                ctx->knn_scratch);
    HIP_TRY(hipGetLastError());
    return SIFT_OK;
}

int sift_knn2_device_noalloc(sift_ctx *ctx, const void *d_q, int nq, const void *d_t, int nt, int train_index_base,
                             void *d_out) {
    if (!ctx || (nq > 0 && (!d_q || !d_out)) || (nt > 0 && !d_t)) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (nq < 0 || nt < 0) return fail(SIFT_ERR_INVALID, "negative count");
    if (nq == 0) return SIFT_OK;
    // two limits, both checked before anything is enqueued: the documented row limit, and the bytes this shape needs
    const size_t need = knn_scratch_bytes(nq, nt);
    if (nt > SIFT_KNN2_NOALLOC_MAX_TRAIN || need > KNN_STATIC_SCRATCH_BYTES)
        return fail(SIFT_ERR_INVALID, "sift_knn2_device_noalloc: " + std::to_string(nq) + " queries x " + std::to_string(nt) +
                                          " train rows need " + std::to_string(need) + " bytes of scratch, the static buffer holds " +
                                          std::to_string(KNN_STATIC_SCRATCH_BYTES) + " (at most " +
                                          std::to_string(SIFT_KNN2_NOALLOC_MAX_TRAIN) + " train rows per call)");
    HIP_TRY(hipSetDevice(ctx->device));
    // every word of the scratch a launch reads is written by that launch first (one bias per train row, one
    // partial record per split and query), so nothing needs clearing between calls
    launch_knn2(ctx->stream, (const uint8_t *)d_q, nq, (const uint8_t *)d_t, nt, train_index_base, (sift_knn2 *)d_out,
                ctx->knn_static);
    HIP_TRY(hipGetLastError());
    return SIFT_OK;
}

int sift_knn2_merge_device(sift_ctx *ctx, const void *d_parts, int nparts, int nq, void *d_out, void *d_accept) {
    if (!ctx || !d_parts) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (nparts < 1 || nq < 0) return fail(SIFT_ERR_INVALID, "bad count");
    HIP_TRY(hipSetDevice(ctx->device));
    launch_knn2_merge(ctx->stream, (const sift_knn2 *)d_parts, nparts, nq, (sift_knn2 *)d_out, (uint8_t *)d_accept);
    HIP_TRY(hipGetLastError());
    return SIFT_OK;
}

int sift_knn2_u8(sift_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, sift_knn2 *out, uint8_t *accept) {
    if (!ctx || (nq > 0 && (!q || !out)) || (nt > 0 && !t)) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (nq < 0 || nt < 0) return fail(SIFT_ERR_INVALID, "negative count");
    if (nq == 0) return SIFT_OK;
    if (any_byte_above_127(q, (size_t)nq * 128)) return fail(SIFT_ERR_NOT_INTEGRAL, "query descriptor value above 127");
    if (any_byte_above_127(t, (size_t)nt * 128)) return fail(SIFT_ERR_NOT_INTEGRAL, "train descriptor value above 127");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t qb = ((size_t)nq * 128 + 255) & ~(size_t)255, tb = ((size_t)nt * 128 + 255) & ~(size_t)255;
    const size_t ob = ((size_t)nq * sizeof(sift_knn2) + 255) & ~(size_t)255;
    const size_t ab = ((size_t)nq + 255) & ~(size_t)255;
    const size_t need = qb + tb + 2 * ob + ab;
    if (need > ctx->knn_io_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        int rc = ensure_bytes(&ctx->knn_io, &ctx->knn_io_bytes, need);
        if (rc) return rc;
    }
    char *base = (char *)ctx->knn_io;
    uint8_t *dq = (uint8_t *)base, *dt = (uint8_t *)(base + qb);
    sift_knn2 *dpart = (sift_knn2 *)(base + qb + tb), *dout = (sift_knn2 *)(base + qb + tb + ob);
    uint8_t *dacc = (uint8_t *)(base + qb + tb + 2 * ob);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dq, q, (size_t)nq * 128, hipMemcpyHostToDevice, st));
    if (nt > 0) HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nt * 128, hipMemcpyHostToDevice, st));
    int rc = sift_knn2_device(ctx, dq, nq, dt, nt, 0, dpart);
    if (rc) return rc;
    rc = sift_knn2_merge_device(ctx, dpart, 1, nq, dout, dacc);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)nq * sizeof(sift_knn2), hipMemcpyDeviceToHost, st));
    if (accept) HIP_TRY(hipMemcpyAsync(accept, dacc, (size_t)nq, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SIFT_OK;
}

int sift_knn2_f32(sift_ctx *ctx, const float *q, int nq, const float *t, int nt, sift_knn2 *out, uint8_t *accept) {
    if (!ctx || (nq > 0 && (!q || !out)) || (nt > 0 && !t)) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (nq < 0 || nt < 0) return fail(SIFT_ERR_INVALID, "negative count");
    std::vector<uint8_t> qb((size_t)nq * 128), tb((size_t)nt * 128);
    auto quantise = [](const float *src, std::vector<uint8_t> &dst) {
        for (size_t i = 0; i < dst.size(); i++) {
            const float v = src[i];
            if (!(v >= 0.0f && v <= 127.0f) || v != std::floor(v)) return false;
            dst[i] = (uint8_t)v;
        }
        return true;
    };
    if (!quantise(q, qb) || !quantise(t, tb))
        return fail(SIFT_ERR_NOT_INTEGRAL, "descriptor values must be integers in [0,127]");
    return sift_knn2_u8(ctx, qb.data(), nq, tb.data(), nt, out, accept);
}

int sift_match_min_dist_filter(const sift_knn2 *knn, int nq, int32_t *good_idx, int32_t *n_good) {
    if ((nq > 0 && (!knn || !good_idx)) || !n_good) return fail(SIFT_ERR_INVALID, "NULL argument");
    double min_dist = 100;  // main.cpp:77
    for (int i = 0; i < nq; i++) {
        if (knn[i].i1 < 0) continue;
        const double dist = std::sqrt((float)knn[i].d1);  // DMatch::distance is float
        if (dist < min_dist) min_dist = dist;
    }
    int g = 0;
    for (int i = 0; i < nq; i++) {
        if (knn[i].i1 < 0) continue;
        const float dist = std::sqrt((float)knn[i].d1);
        if (dist < 3 * min_dist) good_idx[g++] = i;  // main.cpp:88
    }
    *n_good = g;
    return SIFT_OK;
}

// ---- homography -----------------------------------------------------------------

int sift_find_homography(sift_ctx *ctx, const float *src_xy, const float *dst_xy, int n, double reproj_threshold,
                         int max_iters, double confidence, double H[9], uint8_t *mask, sift_homography_info *info) {
    if (!ctx || !H || (n > 0 && (!src_xy || !dst_xy))) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (n < 0) return fail(SIFT_ERR_INVALID, "negative count");
    if (!(confidence > 0 && confidence < 1)) return fail(SIFT_ERR_INVALID, "confidence must lie in (0, 1)");
    if (!(reproj_threshold > 0)) return fail(SIFT_ERR_INVALID, "reprojection threshold must be positive");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t need = homography_workspace_bytes(n, HOMOGRAPHY_CHUNK);
    if (need > ctx->h_work_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        int rc = ensure_bytes(&ctx->h_work, &ctx->h_work_bytes, need);
        if (rc) return rc;
    }
    sift_homography_info local;
    HIP_TRY(find_homography_ransac(ctx->stream, ctx->h_work, src_xy, dst_xy, n, reproj_threshold, max_iters, confidence, H,
                                   mask, info ? info : &local));
    return SIFT_OK;
}

// ---- panorama composition ---------------------------------------------------------

static int stitch_common(sift_ctx *ctx, const uint8_t *ref, int rw, int rh, size_t rstride, const uint8_t *img, int iw, int ih,
                         size_t istride, const double H[9], int dw, int dh, uint8_t *out, int *out_w) {
    if (!ctx || !ref || !H || !out) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (rw < 1 || rh < 1 || dw < 1 || dh < 1 || rstride < (size_t)rw * 3) return fail(SIFT_ERR_INVALID, "bad image size");
    if ((int64_t)dw * dh > (int64_t)1 << 28 || (int64_t)rw * rh > (int64_t)1 << 28) return fail(SIFT_ERR_INVALID, "image too large");
    if (img && (iw < 1 || ih < 1 || istride < (size_t)iw * 3)) return fail(SIFT_ERR_INVALID, "bad image size");
    if (img && ih > rh) return fail(SIFT_ERR_INVALID, "the new image is taller than the panorama");
    double Minv[9];
    if (!invert3x3(H, Minv)) return fail(SIFT_ERR_INVALID, "singular homography");
    HIP_TRY(hipSetDevice(ctx->device));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t ref_b = up((size_t)rw * rh * 3), img_b = img ? up((size_t)iw * ih * 3) : 0, can_b = up((size_t)dw * dh * 3);
    const size_t need = ref_b + img_b + can_b + up((size_t)dw * sizeof(int));
    if (need > ctx->st_work_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        int rc = ensure_bytes(&ctx->st_work, &ctx->st_work_bytes, need);
        if (rc) return rc;
    }
    hipStream_t st = ctx->stream;
    uint8_t *d_ref = (uint8_t *)ctx->st_work, *d_img = d_ref + ref_b, *d_can = d_img + img_b;
    int *d_sums = (int *)(d_can + can_b);
    // images and the canvas cross PCIe through the context's page-locked staging buffer
    HIP_TRY(hipStreamSynchronize(st));
    {
        int rc = ensure_pinned(ctx, ref_b + img_b + can_b);
        if (rc) return rc;
    }
    uint8_t *h_ref = static_cast<uint8_t *>(ctx->pinned), *h_img = h_ref + ref_b, *h_can = h_img + img_b;
    for (int y = 0; y < rh; y++) std::memcpy(h_ref + (size_t)y * rw * 3, ref + rstride * y, (size_t)rw * 3);
    HIP_TRY(hipMemcpyAsync(d_ref, h_ref, (size_t)rw * rh * 3, hipMemcpyHostToDevice, st));
    launch_warp_perspective(st, d_ref, rw, rh, Minv, dw, dh, d_can);
    if (!img) {
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_can, d_can, (size_t)dw * dh * 3, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::memcpy(out, h_can, (size_t)dw * dh * 3);
        return SIFT_OK;
    }
    for (int y = 0; y < ih; y++) std::memcpy(h_img + (size_t)y * iw * 3, img + istride * y, (size_t)iw * 3);
    HIP_TRY(hipMemcpyAsync(d_img, h_img, (size_t)iw * ih * 3, hipMemcpyHostToDevice, st));
    launch_fill_and_blend(st, d_can, dw, d_img, iw, ih);
    launch_column_grey_sums(st, d_can, dw, dh, d_sums);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_can, d_can, (size_t)dw * dh * 3, hipMemcpyDeviceToHost, st));
    std::vector<int> sums((size_t)dw);
    HIP_TRY(hipMemcpyAsync(sums.data(), d_sums, (size_t)dw * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int remove_cols = 0;
    for (int s : sums) remove_cols += s == 0;  // main.cpp:141: every all-black column counts, wherever it lies
    const int nw = dw - remove_cols;
    for (int y = 0; y < dh && nw > 0; y++) std::memcpy(out + (size_t)y * nw * 3, h_can + (size_t)y * dw * 3, (size_t)nw * 3);
    *out_w = nw;
    return SIFT_OK;
}

int sift_warp_perspective_bgr8(sift_ctx *ctx, const uint8_t *src, int sw, int sh, size_t stride, const double H[9], int dw,
                               int dh, uint8_t *dst) {
    return stitch_common(ctx, src, sw, sh, stride, nullptr, 0, 0, 0, H, dw, dh, dst, nullptr);
}

int sift_stitch_pair_bgr8(sift_ctx *ctx, const uint8_t *ref, int ref_w, int ref_h, size_t ref_stride, const uint8_t *img,
                          int img_w, int img_h, size_t img_stride, const double H[9], uint8_t *out, int *out_w) {
    if (!img || !out_w) return fail(SIFT_ERR_INVALID, "NULL argument");
    if (ref_w < 1 || img_w < 1 || (int64_t)ref_w + img_w > INT32_MAX) return fail(SIFT_ERR_INVALID, "bad image size");
    return stitch_common(ctx, ref, ref_w, ref_h, ref_stride, img, img_w, img_h, img_stride, H, ref_w + img_w, ref_h, out, out_w);
}

}  // extern "C"
