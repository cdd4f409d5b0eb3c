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

// Rank of a keypoint = number of keypoints that sort before it.  Candidates
// are first compared on the x key alone, four broadcast LDS words at a time;
// a candidate with the same x (another orientation of the same extremum,
// mostly) is noted in a bit mask, and only those take the full six-word
// comparison afterwards.
constexpr int RS_T = 128;  // threads per workgroup = keys per LDS tile

//
// SPLIT = false: a workgroup ranks its 128 keypoints against the whole list and moves them (grid: tiles x frames).
// SPLIT = true: grid tiles x parts x frames; a workgroup counts only the part of the list that blockIdx.y selects
// and adds its count to ranks[frame][i] (zero on entry), and k_rank_scatter moves the keypoints afterwards: for one
// or a few frames, whose few tiles would otherwise leave most of the chip idle for the length of the list.
template <bool SPLIT>
__global__ __launch_bounds__(RS_T) void k_rank_sort(const DevKeypoint *__restrict__ in, DevKeypoint *__restrict__ out,
                                                     int kp_cap, const FrameCounters *__restrict__ counters,
                                                     int32_t *__restrict__ ranks) {
    __shared__ __attribute__((aligned(16))) uint32_t tile_x[RS_T];
    __shared__ SortKey tile[RS_T];
    const int f = SPLIT ? blockIdx.z : blockIdx.y;
    int n = counters[f].oriented;
    if (n > kp_cap) n = kp_cap;
    const int i0 = blockIdx.x * RS_T;
    if (i0 >= n) return;
    in += (size_t)f * kp_cap;
    out += (size_t)f * kp_cap;
    // this workgroup's share of the list, in whole tiles
    int jlo = 0, jhi = n;
    if (SPLIT) {
        const int tiles = (n + RS_T - 1) / RS_T, per = (tiles + gridDim.y - 1) / gridDim.y;
        jlo = min(n, (int)blockIdx.y * per * RS_T);
        jhi = min(n, ((int)blockIdx.y + 1) * per * RS_T);
        if (jlo >= jhi) return;
    }
    const int i = i0 + threadIdx.x;
    DevKeypoint me;
    SortKey mk, nk;
    for (int q = 0; q < 6; q++) mk.k[q] = nk.k[q] = 0xffffffffu;  // past-the-end keys sort last
    if (i < n) {
        me = in[i];
        mk = make_key(me);
    }
    const uint32_t mx = mk.k[0];
    int rank = 0;
    if (jlo + (int)threadIdx.x < jhi) nk = make_key(in[jlo + threadIdx.x]);
    for (int j0 = jlo; j0 < jhi; j0 += RS_T) {
        tile[threadIdx.x] = nk;
        tile_x[threadIdx.x] = nk.k[0];
        __syncthreads();
        // next tile's key, loaded while this one is scanned
        const int jn = j0 + RS_T + threadIdx.x;
        for (int q = 0; q < 6; q++) nk.k[q] = 0xffffffffu;
        if (jn < jhi) nk = make_key(in[jn]);
        uint32_t eq[RS_T / 32];
#pragma unroll
        for (int w = 0; w < RS_T / 32; w++) eq[w] = 0;
#pragma unroll
        for (int t4 = 0; t4 < RS_T / 4; t4++) {
            const uint4 tx = reinterpret_cast<const uint4 *>(tile_x)[t4];
            rank += (tx.x < mx) + (tx.y < mx) + (tx.z < mx) + (tx.w < mx);
            const uint32_t bits = (tx.x == mx ? 1u : 0u) | (tx.y == mx ? 2u : 0u) | (tx.z == mx ? 4u : 0u) |
                                  (tx.w == mx ? 8u : 0u);
            eq[t4 >> 3] |= bits << ((t4 & 7) * 4);
        }
#pragma unroll
        for (int w = 0; w < RS_T / 32; w++) {
            uint32_t m = eq[w];
            while (m) {
                const int t = w * 32 + __builtin_ctz(m);
                m &= m - 1;
                const int j = j0 + t;
                const int c = cmp_key(tile[t], mk);
                rank += j < jhi && ((c < 0) || (c == 0 && j < i));
            }
        }
        __syncthreads();
    }
    if (i < n) {
        if (SPLIT)
            atomicAdd(&ranks[(size_t)f * kp_cap + i], rank);
        else
            out[rank] = me;
    }
}

__global__ void k_rank_scatter(const DevKeypoint *__restrict__ in, DevKeypoint *__restrict__ out, int kp_cap,
                               const FrameCounters *__restrict__ counters, const int32_t *__restrict__ ranks) {
    const int f = blockIdx.y;
    int n = counters[f].oriented;
    if (n > kp_cap) n = kp_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)f * kp_cap + ranks[(size_t)f * kp_cap + i]] = in[(size_t)f * kp_cap + i];
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
                  uint8_t *flags, int kp_cap, FrameCounters *counters, int32_t *ranks) {
    const int tiles = (kp_cap + RS_T - 1) / RS_T;
    if (ranks) {
        hipMemsetAsync(ranks, 0, (size_t)nframes * kp_cap * sizeof(int32_t), st);
        hipLaunchKernelGGL(k_rank_sort<true>, dim3(tiles, 8, nframes), dim3(RS_T), 0, st, oriented, sorted, kp_cap, counters, ranks);
        hipLaunchKernelGGL(k_rank_scatter, dim3((kp_cap + 255) / 256, nframes), dim3(256), 0, st, oriented, sorted, kp_cap,
                           counters, ranks);
    } else {
        hipLaunchKernelGGL(k_rank_sort<false>, dim3(tiles, nframes), dim3(RS_T), 0, st, oriented, sorted, kp_cap, counters,
                           (int32_t *)nullptr);
    }
    hipLaunchKernelGGL(k_unique_finalize, dim3(nframes), dim3(1024), 0, st, sorted, final_kp, flags, kp_cap, counters);
}

}  // namespace sa
