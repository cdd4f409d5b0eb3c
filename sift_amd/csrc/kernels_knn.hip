// Brute-force top-2 nearest neighbours as an int8 MFMA distance GEMM with the
// top-2 selection fused into the tile loop.
//
// Replaces FlannBasedMatcher::match / knnMatch(k=2) (main.cpp:48-51, 75,
// 186-198) by an exact search.  Descriptors are bytes in [0,127] (the engine
// emits 0..102, sift.cpp:566-573), so squared distances are exact integers.
//
//   d2(q,t) = |q|^2 + |t|^2 - 2 q.t
//           = |q|^2 + [ |t|^2 - C*sum(t) ] + sum_k (C - 2 q_k) t_k,   C = 126
//
// (C - 2 q_k) fits a signed byte for q_k in [0,127], so one
// v_mfma_i32_32x32x32_i8 chain over K = 128 with the bracket ("train bias")
// preloaded into the accumulator yields s = d2 - |q|^2 directly, which ranks
// train rows identically for a fixed query.
//
// Orientation: train rows are the MFMA rows (A operand, streamed through LDS),
// queries are the columns (B operand, held in registers for the whole scan), so
// a lane owns one query and its 16 accumulator registers are 16 train rows.
// The common case costs a 16-way integer min and one compare per 32x32 block:
// only when some lane's minimum beats its current second-best does the block's
// accumulator go through the exact update path.  Ties resolve to the smaller train index.
#include <limits.h>

#include <mutex>

#include "common.h"
#include "kernels.h"

// Scratch of the entry point that allocates nothing (sift_knn2_device_noalloc): train biases and the
// per-split partial results live here instead of in a hipMalloc'ed buffer.  Every GPU the library
// runs on holds its own copy of the array, at its own address, until the process ends.  The name is
// only a host-side handle: host code uses the address hipGetSymbolAddress reports for the current
// device (knn_static_scratch below), never the name itself.
static __device__ __align__(256) unsigned char scratch[sa::KNN_STATIC_SCRATCH_BYTES];

namespace sa {

namespace {


using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;

constexpr int TT = 128;     // train rows per LDS tile
constexpr int QB = 4;       // 32-query column blocks per wave
constexpr int QPB = 4 * QB * 32;  // queries per 256-thread workgroup
constexpr int C_OFF = 126;

// train bias |t|^2 - C*sum(t), one thread per row
__global__ void k_train_bias(const uint8_t *__restrict__ t, int nt, int *__restrict__ bias) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nt) return;
    const uint4 *row = reinterpret_cast<const uint4 *>(t + (size_t)r * 128);
    int sq = 0, sum = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint4 v = row[c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = (w[d] >> (8 * b)) & 0xff;
                sq += x * x;
                sum += x;
            }
    }
    bias[r] = sq - C_OFF * sum;
}

// (s, idx) lexicographic "a before b"
__device__ __forceinline__ bool before(int sa_, int ia, int sb, int ib) {
    return sa_ < sb || (sa_ == sb && (unsigned)ia < (unsigned)ib);  // idx -1 sorts last
}

__device__ __forceinline__ void top2_insert(int s, int idx, int &b1, int &i1, int &b2, int &i2) {
    if (before(s, idx, b1, i1)) {
        b2 = b1; i2 = i1; b1 = s; i1 = idx;
    } else if (before(s, idx, b2, i2)) {
        b2 = s; i2 = idx;
    }
}

// grid.x = query blocks, grid.y = train splits (each split scans a contiguous
// range of tiles and writes its own partial result row).
__global__ __launch_bounds__(256, 2) void k_knn2(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                               int nt, const int *__restrict__ tbias, int train_base,
                                               int tiles_per_split, sift_knn2 *__restrict__ out) {
    __shared__ uint4 s_tile[2][TT * 8];
    __shared__ int s_bias[2][TT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    // ---- query fragments (B operand): lane holds 16 k-values of its query per k-step
    v4i bf[QB][4];
    int qnorm[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const int qi = blockIdx.x * QPB + wv * (QB * 32) + qb * 32 + c;
        int nrm = 0;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (qi < nq) raw = *reinterpret_cast<const uint4 *>(q + (size_t)qi * 128 + ks * 32 + h * 16);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            v4i f;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint32_t o = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int x = (w[d] >> (8 * b)) & 0xff;
                    nrm += x * x;
                    o |= (uint32_t)((C_OFF - 2 * x) & 0xff) << (8 * b);
                }
                f[d] = (int)o;
            }
            bf[qb][ks] = f;
        }
        nrm += __shfl_xor(nrm, 32);
        qnorm[qb] = nrm;
    }

    // Each query is scanned by two lanes (lane and lane ^ 32), each over half of the rows of a
    // block with its own running top-2 (b1 <= b2).  thr is the second smallest of the four
    // values the pair holds: a row at or above it cannot be in the query's final top-2 (an
    // equal distance loses to the smaller index already held), so it gates the update path.
    int b1[QB], b2[QB], i1[QB], i2[QB], thr[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        b1[qb] = b2[qb] = thr[qb] = INT_MAX;
        i1[qb] = i2[qb] = -1;
    }

    const int ntiles = (nt + TT - 1) / TT;
    const int tile_begin = blockIdx.y * tiles_per_split;
    const int tile_end = min(ntiles, tile_begin + tiles_per_split);

    uint4 stage[4];
    int stage_bias = INT_MAX;
    auto fetch = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = tid + 256 * u;
            const int row = tile * TT + (g >> 3);
            stage[u] = make_uint4(0, 0, 0, 0);
            if (row < nt) stage[u] = *reinterpret_cast<const uint4 *>(t + (size_t)row * 128 + (g & 7) * 16);
        }
        if (tid < TT) {
            const int row = tile * TT + tid;
            stage_bias = row < nt ? tbias[row] : INT_MAX;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = tid + 256 * u;
            const int row = g >> 3, chunk = g & 7;
            s_tile[buf][row * 8 + (chunk ^ ((row >> 1) & 7))] = stage[u];
        }
        if (tid < TT) s_bias[buf][tid] = stage_bias;
    };

    if (tile_begin < tile_end) {
        fetch(tile_begin);
        commit(0);
    }
    __syncthreads();

    for (int tile = tile_begin; tile < tile_end; tile++) {
        const int buf = (tile - tile_begin) & 1;
        if (tile + 1 < tile_end) fetch(tile + 1);
        // Pass over the tile.  A chain's accumulator goes through the exact update only when some lane's minimum
        // of its 16 rows beats that lane's threshold (a wave-uniform branch, rarely taken once the thresholds
        // have tightened); the accumulator is still live at that point, so nothing is recomputed.
        //
        // The pass is software-pipelined by hand, one scheduling fence per MFMA: in the shadow of each MFMA the
        // wave issues either two LDS reads of the next block's operands or a third of one chain's 16-way
        // minimum (three vector instructions), and a chain's minimum is only started two MFMAs after the
        // chain's last one, when its result has left the matrix pipe.  MFMA order within a block of four
        // chains c0..c3 (one per 32-query block): c0.1 c1.1 c2.1 | c0.2-4 | c3.1 | c1.2-4 | c2.2-4 | c3.2-4, so that
        // the bias registers (the C operand of each chain's first MFMA) are free for the next block's after the
        // seventh MFMA and chain c3 of the previous block can be reduced under the first six.
        auto load_af = [&](int rb, int ks, v4i &dst) {
            const int row = rb * 32 + c, chunk = ks * 2 + h;
            const uint4 a = s_tile[buf][row * 8 + (chunk ^ ((row >> 1) & 7))];
            dst = v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w};
        };
        auto load_bias = [&](int rb, int g, v16i &cinit) {
            const int4 bb = *reinterpret_cast<const int4 *>(&s_bias[buf][rb * 32 + 8 * g + 4 * h]);
            cinit[4 * g + 0] = bb.x;
            cinit[4 * g + 1] = bb.y;
            cinit[4 * g + 2] = bb.z;
            cinit[4 * g + 3] = bb.w;
        };
        auto half_min = [](const v16i &a, int o, int &x, int &y) {  // three instructions: min of 8 left in two
            const int t = min(min(a[o], a[o + 1]), a[o + 2]);
            x = min(min(a[o + 3], a[o + 4]), a[o + 5]);
            y = min(min(a[o + 6], a[o + 7]), t);
            asm volatile("" : "+v"(x), "+v"(y));  // evaluated here, not where the values are consumed
        };
        // Exact update of query block qb with the 16 rows a lane holds of train block rb (still in the chain's
        // accumulator when its minimum is known), row by row in ascending order (a lane's first of equal
        // distances stays) with a wave vote per row; then the two lanes of each query exchange their best two
        // for the new threshold.  Visiting only the row groups whose partial minimum qualifies, and exchanging
        // through v_permlane32_swap instead of LDS, measured the same (26.1 against 25.8 ms).
        auto exact_update = [&](const v16i &a, int qb, int rb) {
            const int idx0 = tile * TT + rb * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int sv = a[r];
                // most rows lose against the running second-best in every lane: skip them whole
                if (__any(sv < thr[qb])) {
                    const int idx = idx0 + (r & 3) + 8 * (r >> 2);
                    if (sv < b1[qb]) {
                        b2[qb] = b1[qb]; i2[qb] = i1[qb]; b1[qb] = sv; i1[qb] = idx;
                    } else if (sv < b2[qb]) {
                        b2[qb] = sv; i2[qb] = idx;
                    }
                }
            }
            const int p1 = __shfl_xor(b1[qb], 32), p2 = __shfl_xor(b2[qb], 32);
            thr[qb] = min(max(b1[qb], p1), min(b2[qb], p2));
        };
        auto finish = [&](int x1, int y1, int x2, int y2, const v16i &a, int qb, int rb) {
            const int m = min(min(min(x1, y1), x2), y2);
            if (__any(m < thr[qb])) exact_update(a, qb, rb);
        };
        static_assert(QB == 4, "the schedule below is written for four chains");
        constexpr int NB = TT / 32;
        v4i af[4], afn[4];
        v16i cinit;
        v16i acc[QB];
        int lo2[2], hi2[2], lo3[2], hi3[2];  // halves of the minima of chains 2 and 3, carried into the next block
#pragma unroll
        for (int ks = 0; ks < 4; ks++) load_af(0, ks, af[ks]);
#pragma unroll
        for (int g = 0; g < 4; g++) load_bias(0, g, cinit);
#define SA_FENCE() __builtin_amdgcn_sched_barrier(0)
#define SA_MFMA(qb, ks, cin) acc[qb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks], bf[qb][ks], cin, 0, 0, 0)
#pragma unroll
        for (int rb = 0; rb < NB; rb++) {
            const bool prev = rb > 0, more = rb + 1 < NB;
            int lo0[2], hi0[2], lo1[2], hi1[2];
            SA_FENCE();
            SA_MFMA(0, 0, cinit);
            SA_FENCE();
            if (prev) finish(lo2[0], lo2[1], hi2[0], hi2[1], acc[2], 2, rb - 1);
            SA_FENCE();
            SA_MFMA(1, 0, cinit);
            SA_FENCE();
            if (more) {
                load_af(rb + 1, 0, afn[0]);
                load_af(rb + 1, 1, afn[1]);
            }
            SA_FENCE();
            SA_MFMA(2, 0, cinit);
            SA_FENCE();
            if (prev) half_min(acc[3], 0, lo3[0], lo3[1]);
            SA_FENCE();
            SA_MFMA(0, 1, acc[0]);
            SA_FENCE();
            if (prev) half_min(acc[3], 8, hi3[0], hi3[1]);
            SA_FENCE();
            SA_MFMA(0, 2, acc[0]);
            SA_FENCE();
            if (prev) finish(lo3[0], lo3[1], hi3[0], hi3[1], acc[3], 3, rb - 1);
            SA_FENCE();
            SA_MFMA(0, 3, acc[0]);
            SA_FENCE();
            if (more) {
                load_af(rb + 1, 2, afn[2]);
                load_af(rb + 1, 3, afn[3]);
            }
            SA_FENCE();
            SA_MFMA(3, 0, cinit);
            SA_FENCE();
            if (more) {
                load_bias(rb + 1, 0, cinit);
                load_bias(rb + 1, 1, cinit);
            }
            SA_FENCE();
            SA_MFMA(1, 1, acc[1]);
            SA_FENCE();
            if (more) {
                load_bias(rb + 1, 2, cinit);
                load_bias(rb + 1, 3, cinit);
            }
            SA_FENCE();
            SA_MFMA(1, 2, acc[1]);
            SA_FENCE();
            half_min(acc[0], 0, lo0[0], lo0[1]);
            SA_FENCE();
            SA_MFMA(1, 3, acc[1]);
            SA_FENCE();
            half_min(acc[0], 8, hi0[0], hi0[1]);
            SA_FENCE();
            SA_MFMA(2, 1, acc[2]);
            SA_FENCE();
            finish(lo0[0], lo0[1], hi0[0], hi0[1], acc[0], 0, rb);
            SA_FENCE();
            SA_MFMA(2, 2, acc[2]);
            SA_FENCE();
            half_min(acc[1], 0, lo1[0], lo1[1]);
            SA_FENCE();
            SA_MFMA(2, 3, acc[2]);
            SA_FENCE();
            half_min(acc[1], 8, hi1[0], hi1[1]);
            SA_FENCE();
            SA_MFMA(3, 1, acc[3]);
            SA_FENCE();
            finish(lo1[0], lo1[1], hi1[0], hi1[1], acc[1], 1, rb);
            SA_FENCE();
            SA_MFMA(3, 2, acc[3]);
            SA_FENCE();
            half_min(acc[2], 0, lo2[0], lo2[1]);
            SA_FENCE();
            SA_MFMA(3, 3, acc[3]);
            SA_FENCE();
            half_min(acc[2], 8, hi2[0], hi2[1]);
            SA_FENCE();
            if (more) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++) af[ks] = afn[ks];
            }
        }
        finish(lo2[0], lo2[1], hi2[0], hi2[1], acc[2], 2, NB - 1);
        half_min(acc[3], 0, lo3[0], lo3[1]);
        half_min(acc[3], 8, hi3[0], hi3[1]);
        finish(lo3[0], lo3[1], hi3[0], hi3[1], acc[3], 3, NB - 1);
        SA_FENCE();
#undef SA_MFMA
#undef SA_FENCE
        if (tile + 1 < tile_end) commit(buf ^ 1);
        __syncthreads();
    }

    // ---- merge the two lane halves of each query, add |q|^2, store
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const int pb1 = __shfl_xor(b1[qb], 32), pi1 = __shfl_xor(i1[qb], 32);
        const int pb2 = __shfl_xor(b2[qb], 32), pi2 = __shfl_xor(i2[qb], 32);
        int r1 = b1[qb], j1 = i1[qb], r2 = b2[qb], j2 = i2[qb];
        top2_insert(pb1, pi1, r1, j1, r2, j2);
        top2_insert(pb2, pi2, r1, j1, r2, j2);
        const int qi = blockIdx.x * QPB + wv * (QB * 32) + qb * 32 + c;
        if (h == 0 && qi < nq) {
            sift_knn2 o;
            o.d1 = j1 >= 0 ? r1 + qnorm[qb] : INT_MAX;
            o.i1 = j1 >= 0 ? j1 + train_base : -1;
            o.d2 = j2 >= 0 ? r2 + qnorm[qb] : INT_MAX;
            o.i2 = j2 >= 0 ? j2 + train_base : -1;
            out[(size_t)blockIdx.y * nq + qi] = o;
        }
    }
}

// top-2 of nparts partial results per query under (distance, index) order,
// plus the ratio test sqrtf(d1) < 0.7f*sqrtf(d2) (main.cpp:190,194).
__global__ void k_knn2_merge(const sift_knn2 *__restrict__ parts, int nparts, int nq, sift_knn2 *__restrict__ out,
                             uint8_t *__restrict__ accept) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int b1 = INT_MAX, b2 = INT_MAX, i1 = -1, i2 = -1;
    for (int p = 0; p < nparts; p++) {
        const sift_knn2 e = parts[(size_t)p * nq + qi];
        if (e.i1 >= 0) top2_insert(e.d1, e.i1, b1, i1, b2, i2);
        if (e.i2 >= 0) top2_insert(e.d2, e.i2, b1, i1, b2, i2);
    }
    sift_knn2 o = {b1, i1, b2, i2};
    if (out) out[qi] = o;
    if (accept) {
        const float ratio_thresh = 0.7f;
        accept[qi] = (i2 >= 0 && sqrtf((float)b1) < ratio_thresh * sqrtf((float)b2)) ? 1 : 0;
    }
}

int choose_splits(int nq, int nt) {
    const int qblocks = (nq + QPB - 1) / QPB;
    const int ntiles = (nt + TT - 1) / TT;
    int splits = (512 + qblocks - 1) / qblocks;  // at least two workgroups per CU
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    return splits;
}

}  // namespace

// Device address of `scratch` on `device` (which must be the current device), resolved once per device.
hipError_t knn_static_scratch(int device, void **ptr) {
    constexpr int MAX_DEVICES = 64;
    static std::mutex mu;
    static void *resolved[MAX_DEVICES] = {};
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!resolved[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        resolved[device] = p;
    }
    *ptr = resolved[device];
    return hipSuccess;
}

// Train biases, then one partial result row per train split (none when the launch has a single split:
// it writes the caller's output directly).
size_t knn_scratch_bytes(int nq, int nt) {
    const int splits = choose_splits(nq, nt);
    size_t bias = ((size_t)nt * sizeof(int) + 255) & ~(size_t)255;
    return bias + (splits > 1 ? (size_t)splits * nq * sizeof(sift_knn2) : 0);
}

void launch_knn2(hipStream_t st, const uint8_t *q, int nq, const uint8_t *t, int nt, int train_base, sift_knn2 *out,
                 void *scratch) {
    if (nq <= 0) return;
    int *bias = reinterpret_cast<int *>(scratch);
    const size_t bias_bytes = ((size_t)nt * sizeof(int) + 255) & ~(size_t)255;
    sift_knn2 *parts = reinterpret_cast<sift_knn2 *>(reinterpret_cast<char *>(scratch) + bias_bytes);
    if (nt > 0) hipLaunchKernelGGL(k_train_bias, dim3((nt + 255) / 256), dim3(256), 0, st, t, nt, bias);
    const int splits = choose_splits(nq, nt);
    const int ntiles = (nt + TT - 1) / TT;
    const int tps = splits > 0 && ntiles > 0 ? (ntiles + splits - 1) / splits : 1;
    const int qblocks = (nq + QPB - 1) / QPB;
    if (splits == 1) {
        hipLaunchKernelGGL(k_knn2, dim3(qblocks, 1), dim3(256), 0, st, q, nq, t, nt, bias, train_base, tps, out);
    } else {
        hipLaunchKernelGGL(k_knn2, dim3(qblocks, splits), dim3(256), 0, st, q, nq, t, nt, bias, train_base, tps,
                           parts);
        hipLaunchKernelGGL(k_knn2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, parts, splits, nq, out,
                           (uint8_t *)nullptr);
    }
}

void launch_knn2_merge(hipStream_t st, const sift_knn2 *parts, int nparts, int nq, sift_knn2 *out, uint8_t *accept) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_knn2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, parts, nparts, nq, out, accept);
}

}  // namespace sa
