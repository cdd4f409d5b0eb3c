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
// (C - 2 q_k) fits a signed byte for q_k in [0,127], so a chain of two
// v_mfma_i32_16x16x64_i8 over K = 128 with the bracket ("train bias")
// preloaded into the accumulator yields s = d2 - |q|^2 directly, which ranks
// train rows identically for a fixed query.  (The 16x16x64 shape, not 32x32x32:
// under this part's clock management a bare loop of the former sustains 4.1 P op/s
// on random operands, of the latter 3.45, scripts/ubench/mfma_i8_shapes.hip.)
//
// Orientation: train rows are the MFMA rows (A operand, streamed through LDS),
// queries are the columns (B operand, held in registers for the whole scan), so
// a lane owns one query of a 16-query block and its 4 accumulator registers are 4 of the
// block's 16 train rows.  The common case costs a running integer minimum per 16x16 tile and
// one compare and vote per pair of tiles: only when some lane's minimum beats its query's
// current second-best do the pair's accumulators go through the exact update path.  Ties
// resolve to the smaller train index.  The matrix pipe shares the vector issue port and the
// part's power budget with every other instruction of the wave (each vector instruction per
// MFMA takes 9-10 % off the rate of a bare MFMA stream, scripts/ubench/mfma_valu_mix.hip), so
// the loop is built to hold as few of them as it can.
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

constexpr int TT = 256;     // train rows per LDS tile
constexpr int QB = 8;       // 16-query column blocks per wave
// Eight waves share every tile: a workgroup of four re-reads the train set once per 512 queries, and that traffic
// (L2 -> registers -> LDS) costs the matrix pipe a seventh of its rate; with eight it is halved (24.5 -> 22.4 ms at
// 262,144 x 1,048,576).  One such workgroup fills a CU (two waves per SIMD at this register count).
constexpr int KW = 8;        // waves per workgroup
constexpr int KT = 64 * KW;
constexpr int QPB = KW * QB * 16;  // queries per workgroup
constexpr int C_OFF = 126;

// train bias |t|^2 - C*sum(t), one thread per row.  The array runs on to a whole number of tiles with INT_MAX
// (such a row can never beat a threshold), followed by 16 zero words that stand in for the bytes of those rows.
__global__ void k_train_bias(const uint8_t *__restrict__ t, int nt, int *__restrict__ bias, int padded, int *__restrict__ zeros) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < 16) zeros[r] = 0;
    if (r >= nt) {
        if (r < padded) bias[r] = INT_MAX;
        return;
    }
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
__global__ __launch_bounds__(KT, 2) void k_knn2(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                               int nt, const int *__restrict__ tbias, int train_base,
                                               int tiles_per_split, sift_knn2 *__restrict__ out) {
    // one array: tiles, then bias words (16-byte pieces)
    __shared__ uint4 s_all[2 * TT * 8 + 2 * TT / 4];
    uint4(*s_tile)[TT * 8] = reinterpret_cast<uint4(*)[TT * 8]>(s_all);
    int(*s_bias)[TT] = reinterpret_cast<int(*)[TT]>(s_all + 2 * TT * 8);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, c = lane & 15;  // k-group / row-group of the lane, column of its query in a block

    // ---- query fragments (B operand): lane holds 16 k-values of its query per k-step of 64
    v4i bf[QB][2];
    int qnorm[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const int qi = blockIdx.x * QPB + wv * (QB * 16) + qb * 16 + c;
        int nrm = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (qi < nq) raw = *reinterpret_cast<const uint4 *>(q + (size_t)qi * 128 + ks * 64 + g * 16);
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
        nrm += __shfl_xor(nrm, 16);
        nrm += __shfl_xor(nrm, 32);
        qnorm[qb] = nrm;
    }

    // Each query is scanned by four lanes (lane ^ 16, ^ 32, ^ 48), each over a quarter of the rows of a
    // block with its own running top-2 (b1 <= b2).  thr is the second smallest of the eight
    // values the four hold: a row at or above it cannot be in the query's final top-2 (an
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

    // A tile goes from memory straight into the LDS (no register staging, no LDS write instructions): one
    // global_load_lds_dwordx4 of a wave fills 1 KB of consecutive LDS, i.e. 8 rows, so the XOR swizzle of the 16-byte
    // chunks within a row is applied to the SOURCE address of each lane.  Rows past the end of the train set read the
    // zero words behind the bias array (their bias is INT_MAX).  Wave 0 also brings the tile's 256 bias words.
    // The loads are written as inline assembly: for the builtin the compiler assumes the next LDS read may alias the
    // pending write and waits for the load right after issuing it, which would expose every tile's memory latency.
    constexpr int NST = TT * 8 / KT;  // 16-byte pieces of a tile per thread
    using lds_ptr = __attribute__((address_space(3))) void *;
    auto load_to_lds = [](const void *src, unsigned dst) {  // dst: LDS byte address of the wave's 1 KB, wave-uniform
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(dst)
                     : "memory");
    };
    const unsigned lds_tiles = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_ptr)&s_tile[0][wv * 64]);
    const unsigned lds_bias = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_ptr)&s_bias[0][0]);
    const uint8_t *zeros = reinterpret_cast<const uint8_t *>(tbias + ntiles * TT);
    // source of each of this thread's pieces in the next tile to fetch; they move on by one tile per fetch
    const uint8_t *piece[NST];
#pragma unroll
    for (int u = 0; u < NST; u++) {
        const int e = tid + KT * u;
        const int row = e >> 3, chunk = (e & 7) ^ ((row >> 1) & 7);
        piece[u] = t + ((size_t)tile_begin * TT + row) * 128 + chunk * 16;
    }
    const int *bias_piece = tbias + (size_t)tile_begin * TT + 4 * lane;
    auto fetch = [&](int tile, int buf) {
        const unsigned dst = lds_tiles + buf * (unsigned)(TT * 128);
        if (tile == ntiles - 1 && (nt & (TT - 1)) != 0) {  // only the train set's last tile can hold rows past its end
#pragma unroll
            for (int u = 0; u < NST; u++)
                load_to_lds(tile * TT + ((tid + KT * u) >> 3) < nt ? piece[u] : zeros, dst + u * (KT * 16));
        } else {
#pragma unroll
            for (int u = 0; u < NST; u++) load_to_lds(piece[u], dst + u * (KT * 16));
        }
#pragma unroll
        for (int u = 0; u < NST; u++) piece[u] += TT * 128;
        if (wv == 0) load_to_lds(bias_piece, lds_bias + buf * (unsigned)(TT * 4));
        bias_piece += TT;
    };

    if (tile_begin < tile_end) fetch(tile_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int tile = tile_begin; tile < tile_end; tile++) {
        const int buf = (tile - tile_begin) & 1;
        if (tile + 1 < tile_end) fetch(tile + 1, buf ^ 1);
        // Pass over the tile, 16 train rows at a time.  A block of rows is eight chains of two MFMAs, one chain
        // per 16-query block; a chain's accumulator (this lane's four rows of its query) goes through the exact
        // update only when some lane's minimum beats that lane's threshold (a wave-uniform branch, rarely taken
        // once the thresholds have tightened), while it is still live.
        //
        // The pass is software-pipelined by hand, one scheduling fence per MFMA: behind the first MFMA of a
        // chain the wave issues the minimum of the chain two before it (two vector instructions), whose result
        // has left the matrix pipe by then, and one LDS read of the next block's operands (three of the chains);
        // behind the second MFMA the compare and the vote, for the chains of odd row blocks.
        auto load_af = [&](int rb, int ks, v4i &dst) {
            const int row = rb * 16 + c, chunk = ks * 4 + g;
            const uint4 a = s_tile[buf][row * 8 + (chunk ^ ((row >> 1) & 7))];
            dst = v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w};
        };
        auto load_bias = [&](int rb, v4i &dst) {  // rows 4 g .. 4 g + 3 of the block: this lane's accumulator rows
            const int4 bb = *reinterpret_cast<const int4 *>(&s_bias[buf][rb * 16 + 4 * g]);
            dst = v4i{bb.x, bb.y, bb.z, bb.w};
        };
        // Exact update of query block qb with the four rows a lane holds of train block rb, in ascending order
        // (a lane's first of equal distances stays) with a wave vote per row; then the four lanes of each query
        // merge their best two, pair by pair, for the new threshold.
        auto exact_update = [&](const v4i &a, int qb, int rb) {
            int idx0 = tile * TT + rb * 16 + 4 * g;
            asm volatile("" : "+v"(idx0));  // formed here: hoisted above the branch, the row indices cost every tile four instructions
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int sv = a[r];
                if (__any(sv < thr[qb])) {
                    const int idx = idx0 + r;
                    if (sv < b1[qb]) {
                        b2[qb] = b1[qb]; i2[qb] = i1[qb]; b1[qb] = sv; i1[qb] = idx;
                    } else if (sv < b2[qb]) {
                        b2[qb] = sv; i2[qb] = idx;
                    }
                }
            }
            const int p1 = __shfl_xor(b1[qb], 16), p2 = __shfl_xor(b2[qb], 16);
            const int t1 = min(b1[qb], p1), t2 = min(max(b1[qb], p1), min(b2[qb], p2));  // best two of the pair
            const int u1 = __shfl_xor(t1, 32), u2 = __shfl_xor(t2, 32);
            thr[qb] = min(max(t1, u1), min(t2, u2));
        };
        // A chain is looked at two chains later, when its result has long left the matrix pipe.  The branch to the
        // exact update is taken once per four chains and pair of row blocks, on the union of their votes.
        // Row blocks are looked at in pairs: the even block's minimum is only kept (two vector instructions), the odd
        // block's is folded into it (two) and compared with the threshold (one), so a pair costs five instead of six
        // and half the scalar work; both accumulators of a pair stay live until the pair has been settled.
        v4i acc[2][QB];
        int mins[QB];
        unsigned long long vote[QB];
        auto note_min = [&](int qb, int par) {
            int m;
            if (par == 0) m = min(min(min(acc[0][qb][0], acc[0][qb][1]), acc[0][qb][2]), acc[0][qb][3]);
            else m = min(min(min(min(mins[qb], acc[1][qb][0]), acc[1][qb][1]), acc[1][qb][2]), acc[1][qb][3]);
            asm volatile("" : "+v"(m));
            mins[qb] = m;
        };
        // A vote is not read until an MFMA later: a scalar instruction that takes the result of a vector compare
        // straight away stalls the wave, and with it its next MFMA, until the compare has left the vector pipe.
        auto note_vote = [&](int qb) { vote[qb] = __ballot(mins[qb] < thr[qb]); };
        auto settle = [&](int q0, int rb) {  // chains q0 .. q0 + 3 of blocks rb - 1 and rb (rb odd)
            // Marked unlikely, the update code is laid out behind the loop and the common path falls through instead
            // of jumping over it after every group.  The four chains are written out: in code the compiler takes for
            // cold it does not unroll a loop over them, and an accumulator array indexed by a loop variable lives
            // in scratch memory.
            if (__builtin_expect(((vote[q0] | vote[q0 + 1]) | (vote[q0 + 2] | vote[q0 + 3])) != 0, 0)) {
                auto one = [&](int qq) {
                    if (vote[qq] != 0) {
                        exact_update(acc[0][qq], qq, rb - 1);
                        exact_update(acc[1][qq], qq, rb);
                    }
                };
                one(q0);
                one(q0 + 1);
                one(q0 + 2);
                one(q0 + 3);
            }
        };
        constexpr int NB = TT / 16;
        static_assert(QB == 8 && NB % 2 == 0, "the schedule below settles the chains in two groups of four, row blocks in pairs");
        v4i af[2], afn[2], cinit, cinitn;
        load_af(0, 0, af[0]);
        load_af(0, 1, af[1]);
        load_bias(0, cinit);
#define SA_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int rb = 0; rb < NB; rb++) {
            const bool more = rb + 1 < NB;
            const int par = rb & 1;
#pragma unroll
            for (int qb = 0; qb < QB; qb++) {
                const int look = (qb + QB - 2) % QB;            // the chain two back: 6, 7 of the block before, then 0..5
                const int look_par = qb >= 2 ? par : par ^ 1;   // parity of the row block that chain belongs to
                const bool have = qb >= 2 || rb > 0;
                SA_FENCE();
                acc[par][qb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0], bf[qb][0], cinit, 0, 0, 0);
                SA_FENCE();
                if (have) note_min(look, look_par);
                if (more) {
                    if (qb == 2) load_af(rb + 1, 0, afn[0]);
                    if (qb == 3) load_af(rb + 1, 1, afn[1]);
                    if (qb == 4) load_bias(rb + 1, cinitn);
                }
                SA_FENCE();
                acc[par][qb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1], bf[qb][1], acc[par][qb], 0, 0, 0);
                SA_FENCE();
                if (have) {
                    if (look_par == 1) note_vote(look);
                    if (qb == 6 && par == 1) settle(0, rb);                 // chains 0..3 of this pair
                    if (qb == 2 && par == 0 && rb > 0) settle(4, rb - 1);   // chains 4..7 of the pair before, still intact
                }
            }
            SA_FENCE();
            if (more) {
                af[0] = afn[0];
                af[1] = afn[1];
                cinit = cinitn;
            }
        }
        note_min(QB - 2, 1);
        note_vote(QB - 2);
        note_min(QB - 1, 1);
        note_vote(QB - 1);
        settle(4, NB - 1);
        SA_FENCE();
#undef SA_FENCE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's loads into the other buffer have landed
        __syncthreads();
    }

    // ---- merge the four lanes of each query (pair by pair), add |q|^2, store
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        int r1 = b1[qb], j1 = i1[qb], r2 = b2[qb], j2 = i2[qb];
#pragma unroll
        for (int step = 16; step <= 32; step *= 2) {
            const int pb1 = __shfl_xor(r1, step), pi1 = __shfl_xor(j1, step);
            const int pb2 = __shfl_xor(r2, step), pi2 = __shfl_xor(j2, step);
            top2_insert(pb1, pi1, r1, j1, r2, j2);
            top2_insert(pb2, pi2, r1, j1, r2, j2);
        }
        const int qi = blockIdx.x * QPB + wv * (QB * 16) + qb * 16 + c;
        if (g == 0 && qi < nq) {
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
    int splits = (256 + qblocks - 1) / qblocks;  // at least one workgroup per CU
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
static size_t bias_block_bytes(int nt) {  // bias words of whole tiles + the zero words, rounded to 256 bytes
    const size_t padded = (size_t)((nt + TT - 1) / TT) * TT;
    return ((padded + 16) * sizeof(int) + 255) & ~(size_t)255;
}

size_t knn_scratch_bytes(int nq, int nt) {
    const int splits = choose_splits(nq, nt);
    return bias_block_bytes(nt) + (splits > 1 ? (size_t)splits * nq * sizeof(sift_knn2) : 0);
}

void launch_knn2(hipStream_t st, const uint8_t *q, int nq, const uint8_t *t, int nt, int train_base, sift_knn2 *out,
                 void *scratch) {
    if (nq <= 0) return;
    int *bias = reinterpret_cast<int *>(scratch);
    const size_t bias_bytes = bias_block_bytes(nt);
    sift_knn2 *parts = reinterpret_cast<sift_knn2 *>(reinterpret_cast<char *>(scratch) + bias_bytes);
    const int ntiles = (nt + TT - 1) / TT;
    if (nt > 0)
        hipLaunchKernelGGL(k_train_bias, dim3(ntiles * TT / 256), dim3(256), 0, st, t, nt, bias, ntiles * TT,
                           bias + ntiles * TT);
    const int splits = choose_splits(nq, nt);
    const int tps = splits > 0 && ntiles > 0 ? (ntiles + splits - 1) / splits : 1;
    const int qblocks = (nq + QPB - 1) / QPB;
    if (splits == 1) {
        hipLaunchKernelGGL(k_knn2, dim3(qblocks, 1), dim3(KT), 0, st, q, nq, t, nt, bias, train_base, tps, out);
    } else {
        hipLaunchKernelGGL(k_knn2, dim3(qblocks, splits), dim3(KT), 0, st, q, nq, t, nt, bias, train_base, tps,
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
