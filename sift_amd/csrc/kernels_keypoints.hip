// clean_keypoints (sift.cpp:202-225) on the GPU: deterministic sort, the
// reference's std::unique pass, then coordinate halving.
//
// The extrema kernels append keypoints in a nondeterministic order.  Sorting
// uses the reference's comparator (x up, y up, size down, angle up, response
// down, octave down, class_id down), which is a total order up to elements
// that are identical in every field, so the sorted list is deterministic.
// Sorting is a rank sort: each keypoint counts the keypoints that precede it.
//
// std::unique keeps an element iff the predicate of sift.cpp:214-217 says it
// differs from the last KEPT element.  The predicate needs |dx| <= 1e-4, and
// the list is sorted by x, so an element whose x differs from its predecessor
// by more than 1e-4 differs from every earlier element: such elements start
// independent clusters, and each cluster is scanned sequentially by one thread.
#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

struct SortKey {
    uint32_t k[6];
};

// order-preserving map float -> uint32
__device__ __forceinline__ uint32_t f2u(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ SortKey make_key(const DevKeypoint &kp) {
    SortKey s;
    s.k[0] = f2u(kp.x);
    s.k[1] = f2u(kp.y);
    s.k[2] = ~f2u(kp.size);
    s.k[3] = f2u(kp.angle);
    s.k[4] = ~f2u(kp.response);
    s.k[5] = ~((uint32_t)kp.octave ^ 0x80000000u);
    return s;  // class_id is always -1
}

// -1, 0, +1
__device__ __forceinline__ int cmp_key(const SortKey &a, const SortKey &b) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        if (a.k[i] != b.k[i]) return a.k[i] < b.k[i] ? -1 : 1;
    }
    return 0;
}

__global__ __launch_bounds__(256) void k_rank_sort(const DevKeypoint *__restrict__ in, DevKeypoint *__restrict__ out,
                                                    int kp_cap, const FrameCounters *__restrict__ counters) {
    __shared__ SortKey tile[256];
    const int f = blockIdx.y;
    int n = counters[f].oriented;
    if (n > kp_cap) n = kp_cap;
    const int i0 = blockIdx.x * 256;
    if (i0 >= n) return;
    in += (size_t)f * kp_cap;
    out += (size_t)f * kp_cap;
    const int i = i0 + threadIdx.x;
    DevKeypoint me;
    SortKey mk;
    if (i < n) {
        me = in[i];
        mk = make_key(me);
    }
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        if (j < n) tile[threadIdx.x] = make_key(in[j]);
        __syncthreads();
        const int m = min(256, n - j0);
        if (i < n) {
            for (int t = 0; t < m; t++) {
                const int c = cmp_key(tile[t], mk);
                rank += (c < 0) || (c == 0 && (j0 + t) < i);
            }
        }
        __syncthreads();
    }
    if (i < n) out[rank] = me;
}

// predicate of sift.cpp:214-217: true when J differs from I
__device__ __forceinline__ bool differs(const DevKeypoint &I, const DevKeypoint &J) {
    return (double)fabsf(I.x - J.x) > EPS2 || (double)fabsf(I.x - J.x) > EPS2 ||
           (double)fabsf(I.size - J.size) > EPS2 || (double)fabsf(I.angle - J.angle) > EPS2;
}

__global__ __launch_bounds__(1024) void k_unique_finalize(const DevKeypoint *__restrict__ sorted,
                                                           DevKeypoint *__restrict__ final_kp,
                                                           uint8_t *__restrict__ flags, int kp_cap,
                                                           FrameCounters *__restrict__ counters) {
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int f = blockIdx.x;
    int n = counters[f].oriented;
    if (n > kp_cap) n = kp_cap;
    sorted += (size_t)f * kp_cap;
    final_kp += (size_t)f * kp_cap;
    uint8_t *head = flags + (size_t)f * 2 * kp_cap;
    uint8_t *keep = head + kp_cap;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024)
        head[i] = (i == 0) || ((double)fabsf(sorted[i - 1].x - sorted[i].x) > EPS2);
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        if (!head[i]) continue;
        keep[i] = 1;
        int last = i;
        for (int k = i + 1; k < n && !head[k]; k++) {
            const bool d = differs(sorted[last], sorted[k]);
            keep[k] = d;
            if (d) last = k;
        }
    }
    __threadfence_block();
    if (tid == 0) running = 0;
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const int kflag = (i < n) ? keep[i] : 0;
        const unsigned long long m = __ballot(kflag);
        const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int q = 0; q < wv; q++) before += wave_tot[q];
        if (kflag) {
            DevKeypoint k = sorted[i];
            k.x = (float)(k.x * 0.5);  // sift.cpp:220-222
            k.y = (float)(k.y * 0.5);
            k.size = (float)(k.size * 0.5);
            k.octave = (k.octave & ~255) | ((k.octave - 1) & 255);
            final_kp[before + in_wave] = k;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int q = 0; q < 16; q++) t += wave_tot[q];
            running += t;
        }
        __syncthreads();
    }
    if (tid == 0) counters[f].keypoints = running;
}

}  // namespace

void launch_clean(hipStream_t st, int nframes, const DevKeypoint *oriented, DevKeypoint *sorted, DevKeypoint *final_kp,
                  uint8_t *flags, int kp_cap, FrameCounters *counters) {
    dim3 grid((kp_cap + 255) / 256, nframes);
    hipLaunchKernelGGL(k_rank_sort, grid, dim3(256), 0, st, oriented, sorted, kp_cap, counters);
    hipLaunchKernelGGL(k_unique_finalize, dim3(nframes), dim3(1024), 0, st, sorted, final_kp, flags, kp_cap, counters);
}

}  // namespace sa
