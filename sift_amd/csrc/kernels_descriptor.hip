// 4x4x8 descriptors: get_descriptors_parallel::operator() (sift.cpp:464-584).
//
// One workgroup of two waves per keypoint (four or eight in launches of one or a few
// frames); workgroups pull keypoints from a per-frame cursor.  Wave k of n takes
// rows k, k + n, ... of the (2*half_width+1)^2 window.  For a row, the columns that pass the
// rotated-window test of sift.cpp:503 form an interval (two strips
// -1 < s*j + B < 4 intersected), found exactly when the row is set up; the wave sweeps
// only the intervals, four rows side by side with 16 lanes on consecutive columns of
// each.  A sample's gradient magnitude and direction come from the
// precomputed gradient plane of the keypoint's Gaussian level, and its
// Gaussian weight exp(-(i^2+j^2)/(8 w^2)) is the product of two entries of a
// per-keypoint table exp(-t^2/(8 w^2)) (the rotation preserves i^2+j^2), so
// the loop has no transcendental.
//
// The eight trilinear contributions of a sample are scattered with LDS 64-bit
// integer atomics on a fixed-point tensor (see NCOPY and FX_ROW_EXP below).
// Only the inner 4x4x8 bins are kept (the border ring of the 6x6x8 tensor is
// discarded by sift.cpp:557-563, so contributions to it are never issued).  The
// 16 copies are summed, converted back to double, then normalised, clipped at
// 0.2*norm, scaled by 512 and rounded exactly as sift.cpp:564-573 does.  Output
// is one byte per bin (values 0..255; at most 102 in practice).
#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

constexpr int DBINS = 8;
constexpr int INNER = WINDOW_WIDTH * WINDOW_WIDTH * DBINS;  // 128
// Waves per keypoint: two when the launch has keypoints enough to fill the chip (more waves per workgroup cost
// throughput: 16.1 ms against 14.6 per 64 frames with four); four or eight for a launch of one or a few frames, where
// the time of the stage is the time of its largest windows (a 512 x 512 frame: 169 us with two, 92 with four, 75-84
// with eight; a 1080p frame has keypoints enough for four).  The tensor sums are integers, so all three give the same
// descriptor bit for bit.
constexpr int NWAVES_BATCH = 2, NWAVES_SOME = 4, NWAVES_FEW = 8;
// Gradient loads in flight per lane.  With two the kernel needs no register spills and is 6 % slower (per 16 1080p
// frames 3.02 ms against 2.85); fetching a step ahead of the one in hand (twice the registers for what is fetched)
// measured 3.5-3.9 ms at four and at three waves per SIMD.
constexpr int ROWS_IN_FLIGHT = 4;
// The tensor exists in NCOPY copies, and a lane adds to copy (lane & 15).  The LDS takes a 64-bit atomic of a wave in
// four passes of 16 consecutive lanes, and only lanes of one pass can collide: two of them on one address cost 12
// clocks per instruction instead of 7.2, on one bank 8.0, and lanes of different passes never collide
// (scripts/ubench/lds_atomic3.hip).  With 16 copies the lanes of a pass all use different copies, so no two of them
// ever share an address; and with the copies interleaved (word index = bin * NCOPY + copy) a lane's bank is
// 2 * copy + 32 * (bin & 1): different for every lane of a pass, whatever bins they hit.  Every atomic instruction
// therefore runs at the conflict-free rate.  All waves of a workgroup share the 16 copies: integer sums do not depend
// on the order in which the waves' additions arrive.
constexpr int NCOPY = 16;
constexpr int OS = NCOPY;                 // words between orientation bins
constexpr int CS = DBINS * OS;            // between cell columns
constexpr int RS = WINDOW_WIDTH * CS;     // between cell rows

// Intersect [lo, hi] with { j : -1 < s*j + B < WINDOW_WIDTH }; `is` = 1/s.
__device__ __forceinline__ void clip_strip(double s, double is, bool flat, double B, double &lo, double &hi) {
    if (flat) {
        if (!(B > -1 && B < WINDOW_WIDTH)) hi = lo - 4;  // empty
        return;
    }
    const double j1 = (-1 - B) * is, j2 = (WINDOW_WIDTH - B) * is;
    lo = fmax(lo, fmin(j1, j2));
    hi = fmin(hi, fmax(j1, j2));
}

// Contributions are accumulated as unsigned fixed point with LDS 64-bit integer atomics: integer sums do not depend
// on the order of the additions, so every schedule of the kernel gives the same tensor.  The conversion costs
// nothing: the two Gaussian weight factors carry powers of two that put magnitude = mag * w_row * w_col at
// n * 2^-1074 with n the integer nearest mag * w * 2^k, i.e. in the subnormal range, where the bit pattern of a
// double IS the integer n (and stays so up to n < 2^53).  The shares of the trilinear split are products and
// differences of such values, so they are rounded to integers by the multiplier itself and the eight shares of a
// sample add up to n exactly.  A gradient magnitude is below 2^9 and k <= 43, so n < 2^52.
constexpr int FX_ROW_EXP = -500;  // the row weight carries 2^-500 (still a normal number), the column weight the rest

// A wave-uniform double that the vector unit computed, moved to scalar registers.
__device__ __forceinline__ double uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ __forceinline__ double pow2(int e) { return __longlong_as_double((long long)(1023 + e) << 52); }

struct DescShared {
    unsigned long long tensor[INNER * NCOPY];  // after the copies are summed, its first 2*INNER words hold the sums and their squares as doubles
    double wtab[256];
    int idx;
};

// A wave sweeps KS window rows side by side, 64 / KS lanes on consecutive columns of each ("sub-rows"): the column
// intervals of a rotated window are 0 to ~150 samples long, and a whole wave per row leaves a quarter of its lanes
// past the interval's end (measured: 76 % of the lanes of all sweeps carried a sample).  A 64-bit LDS atomic costs the
// same whatever the number of active lanes, so the stage's time follows the number of sweeps.
constexpr int KS = 4, WL = 64 / KS;

// Value of `my` in lane `first + (this lane's sub-row)`: KS wave-uniform reads and KS - 1 selects.
__device__ __forceinline__ int pick_row(int my, int first, int q) {
    int v = __builtin_amdgcn_readlane(my, first);
#pragma unroll
    for (int k = 1; k < KS; k++) {
        const int u = __builtin_amdgcn_readlane(my, first + k);
        v = q >= k ? u : v;
    }
    return v;
}

// COUNT = false is the descriptor kernel.  COUNT = true walks the same samples with the same tests but, instead of
// scattering them, counts them: per frame the samples that contribute (non-zero gradient inside the rotated window)
// and the LDS atomics they would issue (two per cell corner that lies inside the 4x4 grid), into
// work[frame][0..1].  It exists so that the kernel's throughput can be stated in the unit that bounds it.
template <bool COUNT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_descriptors(Pyramid p, const DevKeypoint *__restrict__ kps, int kp_cap,
                                                      const FrameCounters *__restrict__ counters,
                                                      int32_t *__restrict__ cursors, uint8_t *__restrict__ desc,
                                                      unsigned long long *__restrict__ work) {
    constexpr int NT = NWAVES * 64;
    static_assert(NT >= INNER, "one thread per bin in the normalisation");
    // 1024-byte alignment: the byte offset of a cell then has bits 7..9 (the orientation bin) clear
    __shared__ __attribute__((aligned(1024))) DescShared sm;
    const int f = blockIdx.y;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int q = lane / WL, cl = lane & (WL - 1);
    int n = counters[f].keypoints;
    if (n > kp_cap) n = kp_cap;
    unsigned n_samples = 0, n_atomics = 0;  // COUNT only; per lane, per keypoint (a window holds < 2^17 samples)
    // Keypoints are pulled from a per-frame cursor: window areas differ by a factor of 25 between keypoints, and a
    // fixed round-robin deal of keypoints to workgroups measured 4 % slower than the cursor's dynamic balance.
    // The list is sorted by x across all octaves, so concurrent workgroups sample unrelated windows.  Processing
    // the keypoints in (octave, layer, 128-pixel tile) order instead, one eighth of the order per XCD, takes 40 % off
    // the L2 misses and nothing off the time (measured twice, on two generations of this kernel): the kernel does
    // not wait for memory bandwidth, so the list order stays.
    for (;;) {
        if (tid == 0) sm.idx = atomicAdd(&cursors[f], 1);
        for (int t = tid; t < INNER * NCOPY; t += NT) sm.tensor[t] = 0;
        __syncthreads();
        const int it = __builtin_amdgcn_readfirstlane(sm.idx);
        if (it >= n) break;
        const DevKeypoint kpt = kps[(size_t)f * kp_cap + it];
        int octave = kpt.octave & 255;           // sift.cpp:469
        const int layer = (kpt.octave >> 8) & 255;
        if (octave >= 128) octave |= -128;       // sift.cpp:471-473
        const double scale = octave >= 0 ? (1.0 / double(1 << octave)) : (double(1 << -octave));
        const int po = octave + 1;
        const int width = p.w[po], height = p.h[po];
        // gradients of gauss_images[octave + 1][layer] (sift.cpp:476)
        const Grad *image = p.grad[po] + grad_off(p, po, f, layer);
        const int pt_x = __builtin_amdgcn_readfirstlane((int)round(scale * kpt.x));
        const int pt_y = __builtin_amdgcn_readfirstlane((int)round(scale * kpt.y));
        const double angle = uniform(360 - kpt.angle);
        const double cosv = uniform(cos(((double)(angle)*PI) / 180));
        const double sinv = uniform(sin(((double)(angle)*PI) / 180));
        const double weight_multiplier = -0.5 / ((0.5 * WINDOW_WIDTH) * (0.5 * WINDOW_WIDTH));
        const double bins_per_degree = DBINS * 1.0 / 360.0;
        const double hist_width = uniform(SCALE_MULTIPLIER * 0.5 * scale * kpt.size);  // sift.cpp:491
        // The size of a localised keypoint is 1.6 * 2^(layer + offset / 3) * 2^(octave + 1) with layer <= 3 and
        // |offset| < 0.5 (sift.cpp:378), which makes hist_width at most 43.2 in any octave and the half width at most
        // 153: the weight table below has 256 entries, and the bound is enforced rather than assumed.
        const int half_width = __builtin_amdgcn_readfirstlane(
            min(255, (int)fmin(round(hist_width * (WINDOW_WIDTH + 1) / sqrt(2.0)),
                               sqrt((double)height * (double)height + (double)width * (double)width))));  // sift.cpp:492-493
        const double inv_hw = uniform(1.0 / hist_width);
        const double wscale = weight_multiplier * inv_hw * inv_hw;
        // Fixed-point position: fewer than 2^lg samples of less than 2^9 each, so that no sum reaches 2^63.
        const unsigned long long area = (unsigned long long)(2 * (long long)half_width + 1) * (unsigned long long)(2 * (long long)half_width + 1);
        const int fx_bits = min(43, 54 - (64 - __builtin_clzll(area)));
        // table entries: exp(..) * 2^(k - 1074 - FX_ROW_EXP) (column weights); a row weight is a table entry times
        // 2^(FX_ROW_EXP - (k - 1074 - FX_ROW_EXP)) = exp(..) * 2^FX_ROW_EXP
        const double col_scale = pow2(fx_bits - 1074 - FX_ROW_EXP);
        const double tab_to_row = pow2(2 * FX_ROW_EXP + 1074 - fx_bits);
        for (int t = tid; t <= half_width; t += NT) sm.wtab[t] = exp(wscale * (double)(t * t)) * col_scale;
        __syncthreads();
        // bin_row = (sin*j + cos*i)/hist_width + 1.5 = su*j + Bu(i); bin_col = (sin*i - cos*j)/hist_width + 1.5
        // = sv*j + Bv(i) (sift.cpp:497-501); a sample counts when both lie in (-1, 4) (sift.cpp:503).  The loop
        // works with both bins plus one, in (0, 5): truncation is then the floor, and the cell index needs no sign.
        const double su = uniform(sinv * inv_hw), sv = uniform(-cosv * inv_hw);
        const double cu = uniform(cosv * inv_hw), cv = uniform(sinv * inv_hw);
        const bool flat_u = fabs(su) < 1e-12, flat_v = fabs(sv) < 1e-12;
        const double isu = uniform(flat_u ? 0.0 : 1.0 / su), isv = uniform(flat_v ? 0.0 : 1.0 / sv);
        const double half_bins1 = 0.5 * (WINDOW_WIDTH - 1) + 1.0;
        // orientation bin plus 16 (positive for every direction in [-180, 180] and every keypoint angle in [-360, 720])
        const double ang_bins16 = uniform(16.0 - angle * bins_per_degree);
        // byte offset of this lane's copy of cell (-1, -1): a sample's cell offset is added to it
        const int lane_base = (int)offsetof(DescShared, tensor) + (lane & (NCOPY - 1)) * 8 - (WINDOW_WIDTH + 1) * (CS * 8);
        // image bounds on the column: 0 < pt_x + j < width - 1 (sift.cpp:506)
        const int jmin = max(-half_width, 1 - pt_x), jmax = min(half_width, width - 2 - pt_x);
        // Each wave takes rows wv, wv+NWAVES, ...  Row set-up (column interval) is done for 64 rows at once, one row
        // per lane, and handed to the lanes that sweep the row through readlane; ROWS_IN_FLIGHT * KS rows are then
        // swept together so that their gradient loads are all issued before the first one is consumed.
        for (int ib = -half_width + wv; ib <= half_width; ib += NWAVES * 64) {
            int my_jlo = 1 - pt_x, my_jhi = -pt_x;  // empty: its lanes read column 0
            {
                const int i = ib + lane * NWAVES;
                const int win_row = pt_y + i;
                const double my_Bu = cu * (double)i + half_bins1;
                const double my_Bv = cv * (double)i + half_bins1;
                if (i <= half_width && win_row > 0 && win_row < height - 1) {  // sift.cpp:506 (rows)
                    double lo = jmin, hi = jmax;
                    clip_strip(su, isu, flat_u, my_Bu - 1.0, lo, hi);
                    clip_strip(sv, isv, flat_v, my_Bv - 1.0, lo, hi);
                    if (hi >= lo - 2) {
                        my_jlo = max(jmin, (int)floor(lo) - 1);
                        my_jhi = min(jmax, (int)ceil(hi) + 1);
                        // Tighten the estimate to the exact set { j : 0 < bin_row + 1, bin_col + 1 < 5 } (sift.cpp:503), with
                        // the expressions the samples use.  Each of the four conditions is monotone in j (a fused
                        // multiply-add rounds monotonically), so the set is an interval and the sample loop needs no test.
                        auto inside = [&](int j) {
                            const double dj = (double)j;
                            const double br = __builtin_fma(su, dj, my_Bu), bc = __builtin_fma(sv, dj, my_Bv);
                            return br > 0 && bc > 0 && br < WINDOW_WIDTH + 1 && bc < WINDOW_WIDTH + 1;
                        };
                        while (my_jlo <= my_jhi && !inside(my_jlo)) my_jlo++;
                        while (my_jhi >= my_jlo && !inside(my_jhi)) my_jhi--;
                        if (my_jhi < my_jlo) my_jlo = 1 - pt_x, my_jhi = -pt_x;
                    }
                }
            }
            for (int g0 = 0; g0 < 64 && ib + g0 * NWAVES <= half_width; g0 += ROWS_IN_FLIGHT * KS) {
                // per slot r, for the lane's sub-row: pt_x + column interval, first element of the row, bin offsets, row weight
                int xlo[ROWS_IN_FLIGHT], xhi[ROWS_IN_FLIGHT], yyw[ROWS_IN_FLIGHT];
                double Bu[ROWS_IN_FLIGHT], Bv[ROWS_IN_FLIGHT], wi[ROWS_IN_FLIGHT];
                int longest = 0;  // wave-uniform: the longest interval of the group
#pragma unroll
                for (int r = 0; r < ROWS_IN_FLIGHT; r++) {
#pragma unroll
                    for (int k = 0; k < KS; k++)
                        longest = max(longest, __builtin_amdgcn_readlane(my_jhi, g0 + r * KS + k) -
                                                   __builtin_amdgcn_readlane(my_jlo, g0 + r * KS + k) + 1);
                    xlo[r] = pt_x + pick_row(my_jlo, g0 + r * KS, q);
                    xhi[r] = pt_x + pick_row(my_jhi, g0 + r * KS, q);
                    const int i = ib + (g0 + r * KS + q) * NWAVES;
                    const double di = (double)i;
                    Bu[r] = cu * di + half_bins1;  // the expressions of the set-up above
                    Bv[r] = cv * di + half_bins1;
                    wi[r] = sm.wtab[min(abs(i), 255)] * tab_to_row;
                    yyw[r] = min(max(pt_y + i, 0), height - 1) * width;
                }
                for (int c = cl; c - cl < longest; c += WL) {
                    // Everything the rows read (gradients from memory, column weights from the
                    // LDS table) is requested here, before the first atomic of the step: an LDS read
                    // issued between atomics would have to wait for all of them.  The loads are
                    // unconditional, at addresses clamped into the row's interval, so that all are in
                    // flight together; lanes outside the interval discard what they read.
                    Grad g[ROWS_IN_FLIGHT];
                    double wj[ROWS_IN_FLIGHT];
#pragma unroll
                    for (int r = 0; r < ROWS_IN_FLIGHT; r++) {
                        // lanes past the end of their row's interval re-read its last sample instead of the memory behind it
                        g[r] = image[(unsigned)(yyw[r] + min(xlo[r] + c, xhi[r]))];  // sift.cpp:507-510
                        wj[r] = sm.wtab[min(abs(xlo[r] + c - pt_x), 255)];
                    }
                    // keep the compiler from sinking a load into the branch that consumes it
#pragma unroll
                    for (int r = 0; r < ROWS_IN_FLIGHT; r++) asm volatile("" : "+v"(g[r].mag), "+v"(g[r].ori), "+v"(wj[r]));
#pragma unroll
                    for (int r = 0; r < ROWS_IN_FLIGHT; r++) {
                        const int x = xlo[r] + c, j = x - pt_x;
                        // a zero gradient adds nothing to any bin
                        if (x > xhi[r] || g[r].mag == 0.0) continue;
                        const double dj = (double)j;
                        const double bin_row = __builtin_fma(su, dj, Bu[r]);  // in (0, 5)
                        const double bin_col = __builtin_fma(sv, dj, Bv[r]);
                        // Gaussian weight exp(-(i^2+j^2)/(8 w^2)) (sift.cpp:512-517) as a product of two table entries
                        const double magnitude = (g[r].mag * wi[r]) * wj[r];                 // sift.cpp:518, as an integer * 2^-1074
                        const double orientation = __builtin_fma(g[r].ori, bins_per_degree, ang_bins16);  // sift.cpp:519, plus 16
                        const int row_bin = (int)bin_row, col_bin = (int)bin_col, ori_bin = (int)orientation;  // plus 1, 1, 16
                        const double row_pr = __builtin_amdgcn_fract(bin_row), col_pr = __builtin_amdgcn_fract(bin_col);
                        const double ori_pr = __builtin_amdgcn_fract(orientation);
                        // trilinear split (sift.cpp:527-551): each step peels the upper share off the lower one
                        const double m1 = magnitude * row_pr, m0 = magnitude - m1;
                        const double m01 = m0 * col_pr, m00 = m0 - m01;
                        const double m11 = m1 * col_pr, m10 = m1 - m11;
                        // cell (row_bin - 1 + a, col_bin - 1 + b) of the inner 4x4 grid; the border ring is dropped
                        const bool r0 = (unsigned)(row_bin - 1) < (unsigned)WINDOW_WIDTH;
                        const bool r1 = (unsigned)row_bin < (unsigned)WINDOW_WIDTH;
                        const bool c0 = (unsigned)(col_bin - 1) < (unsigned)WINDOW_WIDTH;
                        const bool c1 = (unsigned)col_bin < (unsigned)WINDOW_WIDTH;
                        if constexpr (COUNT) {
                            n_samples += 1;
                            n_atomics += 2 * (((int)r0 + (int)r1) * ((int)c0 + (int)c1));
                            continue;
                        }
                        const int cell = ((row_bin * WINDOW_WIDTH + col_bin) << 10) + lane_base;  // CS * 8 = 1024 bytes per cell
                        const int obits = ori_bin << 7;                                          // OS * 8 = 128 bytes per orientation bin
                        char *const lds = reinterpret_cast<char *>(&sm);
                        unsigned long long *const a0 = reinterpret_cast<unsigned long long *>(lds + (cell | (obits & 0x380)));
                        unsigned long long *const a1 = reinterpret_cast<unsigned long long *>(lds + (cell | ((obits + 128) & 0x380)));
                        if (r0 && c0) {
                            const double hi = m00 * ori_pr;
                            atomicAdd(a0, (unsigned long long)__double_as_longlong(m00 - hi));
                            atomicAdd(a1, (unsigned long long)__double_as_longlong(hi));
                        }
                        if (r0 && c1) {
                            const double hi = m01 * ori_pr;
                            atomicAdd(a0 + CS, (unsigned long long)__double_as_longlong(m01 - hi));
                            atomicAdd(a1 + CS, (unsigned long long)__double_as_longlong(hi));
                        }
                        if (r1 && c0) {
                            const double hi = m10 * ori_pr;
                            atomicAdd(a0 + RS, (unsigned long long)__double_as_longlong(m10 - hi));
                            atomicAdd(a1 + RS, (unsigned long long)__double_as_longlong(hi));
                        }
                        if (r1 && c1) {
                            const double hi = m11 * ori_pr;
                            atomicAdd(a0 + RS + CS, (unsigned long long)__double_as_longlong(m11 - hi));
                            atomicAdd(a1 + RS + CS, (unsigned long long)__double_as_longlong(hi));
                        }
                    }
                }
            }
        }
        if constexpr (COUNT) {
            // wave totals (fewer than 2^32 per keypoint), then one 64-bit add per wave
            for (int sft = 32; sft > 0; sft >>= 1) {
                n_samples += __shfl_xor(n_samples, sft);
                n_atomics += __shfl_xor(n_atomics, sft);
            }
            if (lane == 0) {
                atomicAdd(&work[2 * f], (unsigned long long)n_samples);
                atomicAdd(&work[2 * f + 1], (unsigned long long)n_atomics);
            }
            n_samples = n_atomics = 0;
            __syncthreads();
            continue;
        }
        __syncthreads();
        // inner 4x4x8 bins (sift.cpp:557-563), copies summed in fixed order
        double *vals = reinterpret_cast<double *>(sm.tensor), *red = vals + INNER;
        double v = 0.0;
        if (tid < INNER) {
            // output bin tid = (cell row, cell column, orientation) is tensor bin tid: the copies of a bin are adjacent
            unsigned long long total = 0;
#pragma unroll
            for (int c = 0; c < NCOPY; c++) total += sm.tensor[tid * NCOPY + ((c + tid) & (NCOPY - 1))];
            v = (double)total * pow2(-fx_bits);
        }
        __syncthreads();  // every copy has been read: the tensor's storage is reused below
        if (tid < INNER) {
            vals[tid] = v;
            red[tid] = v * v;
        }
        __syncthreads();
        for (int sft = INNER / 2; sft > 0; sft >>= 1) {
            if (tid < sft) red[tid] += red[tid + sft];
            __syncthreads();
        }
        if (tid < INNER) {
            const double norm = sqrt(red[0]);          // sift.cpp:564-565
            const double thresh = norm * DESCRIPTOR_MAX;  // sift.cpp:566
            double I = vals[tid];
            I = I > thresh ? thresh : I;
            I /= fmax(norm, EPS);
            I = round(I * 512);
            I = I < 0 ? 0 : I;
            I = I > 255 ? 255 : I;
            desc[((size_t)f * kp_cap + it) * 128 + tid] = (uint8_t)I;
        }
        __syncthreads();
    }
}

}  // namespace

template <int NW>
static void launch_descriptors_nw(hipStream_t st, const Pyramid &p, int nframes, const DevKeypoint *kps, int kp_cap,
                                  const FrameCounters *counters, int32_t *cursors, uint8_t *desc, int blocks) {
    hipLaunchKernelGGL((k_descriptors<false, NW>), dim3(blocks, nframes), dim3(NW * 64), 0, st, p, kps,
                       kp_cap, counters, cursors, desc, (unsigned long long *)nullptr);
}

void launch_descriptors(hipStream_t st, const Pyramid &p, int nframes, const DevKeypoint *kps, int kp_cap,
                        const FrameCounters *counters, int32_t *cursors, uint8_t *desc, int64_t launch_pixels) {
    if (launch_pixels <= ((int64_t)1 << 20))
        launch_descriptors_nw<NWAVES_FEW>(st, p, nframes, kps, kp_cap, counters, cursors, desc, 512);  // two workgroups per CU
    else if (launch_pixels <= ((int64_t)1 << 22))
        launch_descriptors_nw<NWAVES_SOME>(st, p, nframes, kps, kp_cap, counters, cursors, desc, 1024);
    else
        launch_descriptors_nw<NWAVES_BATCH>(st, p, nframes, kps, kp_cap, counters, cursors, desc, 2048 / NWAVES_BATCH);
}

void launch_descriptor_work(hipStream_t st, const Pyramid &p, int nframes, const DevKeypoint *kps, int kp_cap,
                            const FrameCounters *counters, int32_t *cursors, unsigned long long *work) {
    dim3 grid(2048 / NWAVES_BATCH, nframes);
    hipLaunchKernelGGL((k_descriptors<true, NWAVES_BATCH>), grid, dim3(NWAVES_BATCH * 64), 0, st, p, kps, kp_cap, counters,
                       cursors, (uint8_t *)nullptr, work);
}

}  // namespace sa
