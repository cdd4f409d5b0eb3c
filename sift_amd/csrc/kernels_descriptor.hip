// 4x4x8 descriptors: get_descriptors_parallel::operator() (sift.cpp:464-584).
//
// One 256-thread workgroup per keypoint; workgroups pull keypoints from a
// per-frame cursor.  Each wave takes every fourth row of the
// (2*half_width+1)^2 window.  For a row, the columns that can pass the
// rotated-window test of sift.cpp:503 form an interval (two strips |a*j+b| < L
// intersected); the wave sweeps only that interval, widened by one pixel, with
// lanes on consecutive columns (coalesced reads), and every sample still takes
// the exact test.  A sample's gradient magnitude and direction come from the
// precomputed gradient plane of the keypoint's Gaussian level, and its
// Gaussian weight exp(-(i^2+j^2)/(8 w^2)) is the product of two entries of a
// per-keypoint table exp(-t^2/(8 w^2)) (the rotation preserves i^2+j^2), so
// the loop has no transcendental.
//
// The eight trilinear contributions of a sample are scattered with LDS double
// atomics.  Only the inner 4x4x8 bins are kept (the border ring of the 6x6x8
// tensor is discarded by sift.cpp:557-563, so contributions to it are never
// issued), and there is one 128-bin copy per (wave, lane & 7) so that
// neighbouring pixels, which mostly share a bin, rarely collide.  The 32
// copies are summed in fixed order, then normalised, clipped at 0.2*norm,
// scaled by 512 and rounded exactly as sift.cpp:564-573 does.  Output is one
// byte per bin (values 0..255; at most 102 in practice).
#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

constexpr int DBINS = 8;
constexpr int INNER = WINDOW_WIDTH * WINDOW_WIDTH * DBINS;  // 128
constexpr int NWAVES = 4;
constexpr int COPIES_PER_WAVE = 8;
constexpr int NT = NWAVES * 64;  // threads per workgroup
constexpr int ROWS_IN_FLIGHT = 4;
constexpr int NCOPY = NWAVES * COPIES_PER_WAVE;
constexpr int CSTRIDE = INNER + 1;  // copies start one LDS bank pair apart, so equal bins of different copies do not share a bank

// Intersect [lo, hi] with { j : |a*j + b| < L }.
__device__ __forceinline__ void clip_strip(double a, double b, double L, double &lo, double &hi) {
    if (fabs(a) < 1e-12) {
        if (!(fabs(b) < L)) hi = lo - 4;  // empty
        return;
    }
    double j1 = (-L - b) / a, j2 = (L - b) / a;
    if (j1 > j2) {
        const double t = j1;
        j1 = j2;
        j2 = t;
    }
    lo = fmax(lo, j1);
    hi = fmin(hi, j2);
}

__global__ __launch_bounds__(NT) void k_descriptors(Pyramid p, const DevKeypoint *__restrict__ kps, int kp_cap,
                                                      const FrameCounters *__restrict__ counters,
                                                      int32_t *__restrict__ cursors, uint8_t *__restrict__ desc) {
    __shared__ double tensor[NCOPY][CSTRIDE];
    __shared__ double vals[INNER];
    __shared__ double red[INNER];
    __shared__ double wtab[256];
    __shared__ int s_idx;
    const int f = blockIdx.y;
    const int tid = threadIdx.x;
    const int wv = tid >> 6, lane = tid & 63;
    int n = counters[f].keypoints;
    if (n > kp_cap) n = kp_cap;
    for (;;) {
        if (tid == 0) s_idx = atomicAdd(&cursors[f], 1);
        for (int t = tid; t < NCOPY * CSTRIDE; t += NT) (&tensor[0][0])[t] = 0.0;
        __syncthreads();
        const int it = s_idx;
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
        const int pt_x = (int)round(scale * kpt.x);
        const int pt_y = (int)round(scale * kpt.y);
        const double angle = 360 - kpt.angle;
        const double cosv = cos(((double)(angle)*PI) / 180);
        const double sinv = sin(((double)(angle)*PI) / 180);
        const double weight_multiplier = -0.5 / ((0.5 * WINDOW_WIDTH) * (0.5 * WINDOW_WIDTH));
        const double bins_per_degree = DBINS * 1.0 / 360.0;
        const double hist_width = SCALE_MULTIPLIER * 0.5 * scale * kpt.size;  // sift.cpp:491
        const int half_width =
            (int)fmin(round(hist_width * (WINDOW_WIDTH + 1) / sqrt(2.0)),
                      sqrt((double)height * (double)height + (double)width * (double)width));  // sift.cpp:492-493
        const double inv_hw = 1.0 / hist_width;
        const double wscale = weight_multiplier * inv_hw * inv_hw;
        const bool use_tab = half_width < 256;
        if (use_tab)
            for (int t = tid; t <= half_width; t += NT) wtab[t] = exp(wscale * (double)(t * t));
        __syncthreads();
        double *T = tensor[wv * COPIES_PER_WAVE + (lane & (COPIES_PER_WAVE - 1))];
        // bin in (-1, 4)  <=>  |rot / hist_width| < 2.5
        const double L = 0.5 * (WINDOW_WIDTH + 1) * hist_width;
        // image bounds on the column: 0 < pt_x + j < width - 1 (sift.cpp:506)
        const int jmin = max(-half_width, 1 - pt_x), jmax = min(half_width, width - 2 - pt_x);
        // Each wave takes rows wv, wv+4, ...; ROWS_IN_FLIGHT rows are handled together so
        // that their gradient loads are all issued before the first one is consumed.
        for (int i0 = -half_width + wv; i0 <= half_width; i0 += NWAVES * ROWS_IN_FLIGHT) {
            int jlo[ROWS_IN_FLIGHT], jhi[ROWS_IN_FLIGHT];
            int longest = 0;
#pragma unroll
            for (int r = 0; r < ROWS_IN_FLIGHT; r++) {
                const int i = i0 + r * NWAVES;
                const int win_row = pt_y + i;
                jlo[r] = 0;
                jhi[r] = -1;
                if (i <= half_width && win_row > 0 && win_row < height - 1) {  // sift.cpp:506 (rows)
                    double lo = jmin, hi = jmax;
                    clip_strip(sinv, cosv * i, L, lo, hi);    // |sin*j + cos*i| < L
                    clip_strip(-cosv, sinv * i, L, lo, hi);   // |sin*i - cos*j| < L
                    if (hi >= lo - 2) {
                        jlo[r] = max(jmin, (int)floor(lo) - 1);
                        jhi[r] = min(jmax, (int)ceil(hi) + 1);
                    }
                }
                longest = max(longest, jhi[r] - jlo[r] + 1);
            }
            for (int c = 0; c < longest; c += 64) {
                Grad g[ROWS_IN_FLIGHT];
#pragma unroll
                for (int r = 0; r < ROWS_IN_FLIGHT; r++) {
                    const int j = jlo[r] + c + lane;
                    g[r].mag = 0.0;
                    g[r].ori = 0.0;
                    if (j <= jhi[r]) g[r] = image[(size_t)(pt_y + i0 + r * NWAVES) * width + pt_x + j];  // sift.cpp:507-510
                }
#pragma unroll
                for (int r = 0; r < ROWS_IN_FLIGHT; r++) {
                    const int i = i0 + r * NWAVES;
                    const int j = jlo[r] + c + lane;
                    if (j > jhi[r]) continue;
                    const double row_rotation = sinv * j + cosv * i;  // sift.cpp:497
                    const double col_rotation = sinv * i - cosv * j;  // sift.cpp:498
                    const double rr = row_rotation * inv_hw, cr = col_rotation * inv_hw;
                    const double bin_row = rr + 0.5 * (WINDOW_WIDTH - 1);
                    const double bin_col = cr + 0.5 * (WINDOW_WIDTH - 1);
                    if (!(bin_row > -1 && bin_col > -1 && bin_row < WINDOW_WIDTH && bin_col < WINDOW_WIDTH)) continue;
                    const double weight =
                        use_tab ? wtab[abs(i)] * wtab[abs(j)] : exp(wscale * (double)(i * i + j * j));
                    const double magnitude = g[r].mag * weight;                       // sift.cpp:512-518
                    const double orientation = (g[r].ori - angle) * bins_per_degree;  // sift.cpp:519
                    const int row_bin = (int)floor(bin_row);
                    const int col_bin = (int)floor(bin_col);
                    int orient_bin = (int)floor(orientation);
                    const double row_bin_pr = bin_row - row_bin;
                    const double col_bin_pr = bin_col - col_bin;
                    const double orient_bin_pr = orientation - orient_bin;
                    if (orient_bin < 0) orient_bin += DBINS;
                    if (orient_bin >= DBINS) orient_bin -= DBINS;
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int b = 0; b < 2; b++) {
                            // tensor cell (row_bin+1+a, col_bin+1+b) of the 6x6 grid; inner cells are 1..4
                            const int ri = row_bin + a, cj = col_bin + b;
                            if ((unsigned)ri < (unsigned)WINDOW_WIDTH && (unsigned)cj < (unsigned)WINDOW_WIDTH) {
#pragma unroll
                                for (int k = 0; k < 2; k++) {
                                    double cc = magnitude;
                                    cc *= (!a ? (1 - row_bin_pr) : (row_bin_pr));
                                    cc *= (!b ? (1 - col_bin_pr) : (col_bin_pr));
                                    cc *= (!k ? (1 - orient_bin_pr) : (orient_bin_pr));
                                    const int ob = ((orient_bin + k) % DBINS + DBINS) % DBINS;
                                    atomicAdd(&T[(ri * WINDOW_WIDTH + cj) * DBINS + ob], cc);
                                }
                            }
                        }
                }
            }
        }
        __syncthreads();
        // inner 4x4x8 bins (sift.cpp:557-563), copies summed in fixed order
        if (tid < INNER) {
            double v = tensor[0][tid];
#pragma unroll 8
            for (int c = 1; c < NCOPY; c++) v += tensor[c][tid];
            vals[tid] = v;
            red[tid] = v * v;
        }
        __syncthreads();
        for (int sft = INNER / 2; sft > 0; sft >>= 1) {
            if (tid < sft) red[tid] += red[tid + sft];
            __syncthreads();
        }
        if (tid < INNER) {
            const double norm = sqrt(red[0]);            // sift.cpp:564-565
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

void launch_descriptors(hipStream_t st, const Pyramid &p, int nframes, const DevKeypoint *kps, int kp_cap,
                        const FrameCounters *counters, int32_t *cursors, uint8_t *desc) {
    dim3 grid(2048 / NWAVES, nframes);
    hipLaunchKernelGGL(k_descriptors, grid, dim3(NT), 0, st, p, kps, kp_cap, counters, cursors, desc);
}

}  // namespace sa
