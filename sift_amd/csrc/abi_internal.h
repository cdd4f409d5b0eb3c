// Internals shared by the translation units that implement the C ABI (contexts, batches, error reporting).
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace sa {

// Records `msg` as the calling thread's last error and returns `code`.
int fail(int code, const std::string &msg);

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return sa::fail(SIFT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

// Grow-only device buffer: frees and reallocates when `want` exceeds `*have`.
int ensure_bytes(void **buf, size_t *have, size_t want);

// True when some byte of [p, p + n) has its top bit set (a descriptor value above 127): eight bytes per step.
inline bool any_byte_above_127(const uint8_t *p, size_t n) {
    size_t i = 0;
    uint64_t acc = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(p + i) & 7); i++) acc |= p[i];
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        acc |= w;
    }
    for (; i < n; i++) acc |= p[i];
    return (acc & 0x8080808080808080ull) != 0;
}

}  // namespace sa

struct sift_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void *knn_scratch = nullptr;
    size_t knn_scratch_bytes = 0;
    void *knn_static = nullptr;  // this device's copy of the matcher's file-scope scratch array (sift_knn2_device_noalloc)
    void *knn_io = nullptr;  // device staging of the host-buffer matcher calls
    size_t knn_io_bytes = 0;
    void *h_work = nullptr;  // device workspace of sift_find_homography
    size_t h_work_bytes = 0;
    void *st_work = nullptr;  // device workspace of the composition calls
    size_t st_work_bytes = 0;
    hipStream_t aux = nullptr, aux2 = nullptr;   // side streams of the small-workload schedule (run_pipeline): even, odd octaves
    hipEvent_t oct_ready[sa::MAX_OCTAVES] = {};  // level 3 of octave o is complete
    hipEvent_t aux_done = nullptr, aux2_done = nullptr;
    hipEvent_t group_ev[5] = {};  // timing events at the reference's stage boundaries (run_pipeline's `group_ms`)
    void *pinned = nullptr;  // page-locked host staging of the single-image calls and of result downloads
    size_t pinned_bytes = 0;
    sift_batch *cached = nullptr;  // workspace of the single-image calls
    int single_cap = 0;            // its initial per-frame list capacity (0: the batch default)
    long long pipeline_runs = 0;   // launches of the extraction pipeline on this context (sift_ctx_pipeline_runs)
};

struct sift_batch {
    sift_ctx *ctx = nullptr;
    int w = 0, h = 0, max_frames = 0;
    sa::Pyramid pyr{};
    uint8_t *d_frames = nullptr;
    const uint8_t *d_ext_frames = nullptr;  // caller-owned frames, if set
    uint8_t *d_bgr = nullptr;  // one frame of BGR staging, allocated on first use
    sa::Candidate *d_cand = nullptr;
    int cand_cap = 0;
    sa::DevKeypoint *d_oriented = nullptr, *d_sorted = nullptr, *d_final = nullptr;
    uint8_t *d_flags = nullptr;
    int kp_cap = 0;
    uint8_t *d_desc = nullptr;
    sa::FrameCounters *d_counters = nullptr;
    unsigned long long *d_work = nullptr;  // [max_frames][2], sift_batch_descriptor_work only
    int32_t *d_cursors = nullptr;  // [3][max_frames]: (unused), descriptor cursor, localised count
    sa::Taps base_taps{};
    int base_ntaps = 0;
    sa::Taps lvl_taps[sa::IMAGES]{};
    int lvl_ntaps[sa::IMAGES]{};
    std::vector<hipEvent_t> events;
};

namespace sa {

// exec() (sift.cpp:67-98) for frames [0, n) of a batch as a sequence of launches on the context stream;
// `times` (may be null) receives per-kernel-stage device times and serialises the run.  `group_ms` (may be null;
// ignored with `times`) makes the run record five events at the boundaries of the reference's TIMEIT stages
// (sift.cpp:71,72,86,87,92) without changing the schedule; read them with pipeline_group_times after the stream
// has been synchronised.
int run_pipeline(sift_batch *b, int n, sift_stage_times *times, bool group_events = false);
int pipeline_group_times(sift_ctx *ctx, double ms[5]);

}  // namespace sa
