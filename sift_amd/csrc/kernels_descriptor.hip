// 4x4x8 descriptors: get_descriptors_parallel::operator() (sift.cpp:464-584).
//
// One 256-thread workgroup per keypoint; workgroups pull keypoints from a
// per-frame cursor.  Threads sweep the (2*half_width+1)^2 window row-major (so
// image reads are coalesced), and scatter each accepted sample's eight
// trilinear contributions into a per-wave copy of the 6x6x8 tensor in LDS with
// LDS double atomics.  The four copies are then summed in fixed order, the
// inner 4x4x8 bins normalised, clipped at 0.2*norm, scaled by 512 and rounded
// exactly as sift.cpp:557-573 does.  Output is one byte per bin (values 0..255;
// at most 102 in practice).
#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

constexpr int DW = WINDOW_WIDTH + 2;       // 6
constexpr int DBINS = 8;
constexpr int TENSOR = DW * DW * DBINS;    // 288
constexpr int NWAVES = 4;

__global__ __launch_bounds__(256) void k_descriptors(Pyramid p, const DevKeypoint *__restrict__ kps, int kp_cap,
                                                      const FrameCounters *__restrict__ counters,
                                                      int32_t *__restrict__ cursors, uint8_t *__restrict__ desc) {
    __shared__ double tensor[NWAVES][TENSOR];
    __shared__ double vals[128];
    __shared__ double red[128];
    __shared__ int s_idx;
    const int f = blockIdx.y;
    const int tid = threadIdx.x;
    const int wv = tid >> 6;
    int n = counters[f].keypoints;
    if (n > kp_cap) n = kp_cap;
    for (;;) {
        if (tid == 0) s_idx = atomicAdd(&cursors[f], 1);
        for (int t = tid; t < NWAVES * TENSOR; t += 256) (&tensor[0][0])[t] = 0.0;
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
        const double *image = p.gauss[po] + gauss_off(p, po, f, layer);  // sift.cpp:476
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
        const int n1 = 2 * half_width + 1;
        const int total = n1 * n1;
        const float inv_n1 = 1.0f / (float)n1;
        double *T = tensor[wv];
        for (int s = tid; s < total; s += 256) {
            int qi = (int)((float)s * inv_n1);
            int rj = s - qi * n1;
            if (rj < 0) { qi--; rj += n1; }
            if (rj >= n1) { qi++; rj -= n1; }
            const int i = qi - half_width, j = rj - half_width;
            const double row_rotation = sinv * j + cosv * i;  // sift.cpp:497
            const double col_rotation = sinv * i - cosv * j;  // sift.cpp:498
            const double rr = row_rotation / hist_width, cr = col_rotation / hist_width;
            const double bin_row = rr + 0.5 * (WINDOW_WIDTH - 1);
            const double bin_col = cr + 0.5 * (WINDOW_WIDTH - 1);
            if (!(bin_row > -1 && bin_col > -1 && bin_row < WINDOW_WIDTH && bin_col < WINDOW_WIDTH)) continue;
            const int win_row = pt_y + i, win_col = pt_x + j;
            if (!(win_row > 0 && win_col > 0 && win_row < height - 1 && win_col < width - 1)) continue;
            const double *q = image + (size_t)win_row * width + win_col;
            const double dx = q[1] - q[-1];
            const double dy = q[-width] - q[width];
            const double mag = sqrt(dx * dx + dy * dy);
            const double orient = (atan2(dy, dx) * 180) / PI;  // |.| <= 180, so fmod(.,360) is the identity
            const double exponent = rr * rr + cr * cr;
            const double weight = exp(weight_multiplier * exponent);
            const double magnitude = mag * weight;
            const double orientation = (orient - angle) * bins_per_degree;  // sift.cpp:519
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
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        double c = magnitude;
                        c *= (!a ? (1 - row_bin_pr) : (row_bin_pr));
                        c *= (!b ? (1 - col_bin_pr) : (col_bin_pr));
                        c *= (!k ? (1 - orient_bin_pr) : (orient_bin_pr));
                        const int ob = ((orient_bin + k) % DBINS + DBINS) % DBINS;
                        atomicAdd(&T[((row_bin + 1 + a) * DW + (col_bin + 1 + b)) * DBINS + ob], c);
                    }
        }
        __syncthreads();
        // inner 4x4x8 bins (sift.cpp:557-563), wave copies summed in fixed order
        if (tid < 128) {
            const int i = tid >> 5, j = (tid >> 3) & 3, k = tid & 7;
            const int o = ((i + 1) * DW + (j + 1)) * DBINS + k;
            double v = tensor[0][o];
#pragma unroll
            for (int c = 1; c < NWAVES; c++) v += tensor[c][o];
            vals[tid] = v;
            red[tid] = v * v;
        }
        __syncthreads();
        for (int sft = 64; sft > 0; sft >>= 1) {
            if (tid < sft) red[tid] += red[tid + sft];
            __syncthreads();
        }
        if (tid < 128) {
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
    dim3 grid(512, nframes);
    hipLaunchKernelGGL(k_descriptors, grid, dim3(256), 0, st, p, kps, kp_cap, counters, cursors, desc);
}

}  // namespace sa
