// Panorama composition, the tail of the reference's main (main.cpp:115-145):
// warpPerspective of the running panorama into a canvas as wide as both
// images, per-channel hole filling from the new image, the 0.7 / 0.3 blend over
// the new image's rectangle, and the per-column grey sums that decide the crop.
//
// warpPerspective follows OpenCV's INTER_LINEAR / BORDER_CONSTANT path
// (imgproc imgwarp.cpp): the inverse matrix by the 3x3 closed form, source
// coordinates in 1/32 pixel from a per-block origin (64 x 16 blocks: the double
// arithmetic depends on how a column splits into block origin + offset), 15-bit
// integer weights, black outside the source.  One thread per canvas pixel; the
// source is read through the caches (a gather of four 3-byte neighbours).
#include <climits>

#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

struct Mat3 {
    double m[9];
};

__device__ __forceinline__ int clamp_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

__global__ __launch_bounds__(256) void k_warp_perspective(const uint8_t *__restrict__ src, int sw, int sh, Mat3 M, int dw, int dh,
                                                          int bw0, uint8_t *__restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int xb = x / bw0 * bw0, x1 = x - xb;
    const double X0 = M.m[0] * xb + M.m[1] * y + M.m[2];
    const double Y0 = M.m[3] * xb + M.m[4] * y + M.m[5];
    const double W0 = M.m[6] * xb + M.m[7] * y + M.m[8];
    double W = W0 + M.m[6] * x1;
    W = W != 0.0 ? 32 / W : 0;
    const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + M.m[0] * x1) * W));
    const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + M.m[3] * x1) * W));
    const int X = (int)rint(fX), Y = (int)rint(fY);  // saturate_cast<int>: round half to even
    const int sx = max(SHRT_MIN, min(SHRT_MAX, X >> 5)), sy = max(SHRT_MIN, min(SHRT_MAX, Y >> 5));
    uint8_t *D = dst + ((size_t)y * dw + x) * 3;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        D[0] = D[1] = D[2] = 0;
        return;
    }
    // weights of the 32 x 32 sub-pixel table: (32-fx)(32-fy)/1024 in 15 bits; its (0, 0) entry saturates to 32767
    // and carries the missing 1 on its last weight
    const int fx = X & 31, fy = Y & 31;
    int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    if (w0 == 32768) w0 = 32767, w3 = 1;
    const bool x0in = sx >= 0, x1in = sx + 1 < sw, y0in = sy >= 0, y1in = sy + 1 < sh;
    const uint8_t *p00 = src + ((size_t)max(sy, 0) * sw + max(sx, 0)) * 3;
    const uint8_t *p01 = src + ((size_t)max(sy, 0) * sw + min(sx + 1, sw - 1)) * 3;
    const uint8_t *p10 = src + ((size_t)min(sy + 1, sh - 1) * sw + max(sx, 0)) * 3;
    const uint8_t *p11 = src + ((size_t)min(sy + 1, sh - 1) * sw + min(sx + 1, sw - 1)) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int v0 = x0in && y0in ? p00[k] : 0, v1 = x1in && y0in ? p01[k] : 0;
        const int v2 = x0in && y1in ? p10[k] : 0, v3 = x1in && y1in ? p11[k] : 0;
        D[k] = (uint8_t)clamp_u8((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
    }
}

// main.cpp:120-134 over the new image's rectangle of the canvas: a zero channel takes the image's value, then
// canvas = saturate(0.7f * image + 0.3f * canvas) in float, rounded half to even.
__global__ __launch_bounds__(256) void k_fill_and_blend(uint8_t *__restrict__ canvas, int cw, const uint8_t *__restrict__ img, int iw,
                                                        int ih) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // channel sample of the image
    if (i >= (size_t)iw * ih * 3) return;
    const size_t row = i / ((size_t)iw * 3), col = i - row * iw * 3;
    uint8_t *p = canvas + row * cw * 3 + col;
    const uint8_t q = img[i];
    const uint8_t c = *p == 0 ? q : *p;
    const float v = (float)q * 0.7f + (float)c * 0.3f + 0.f;
    *p = (uint8_t)clamp_u8((int)rintf(v));
}

// Per column, the sum over the rows of the 15-bit BGR -> grey value (main.cpp:137-140).
__global__ __launch_bounds__(256) void k_column_grey_sums(const uint8_t *__restrict__ canvas, int cw, int ch, int *__restrict__ sums) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= cw) return;
    int s = 0;
    for (int y = 0; y < ch; y++) {
        const uint8_t *p = canvas + ((size_t)y * cw + x) * 3;
        s += (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
    }
    sums[x] = s;
}

}  // namespace

// cv::invert of a 3x3 double matrix (core lapack.cpp, the closed form used for n = 3); false when singular.
bool invert3x3(const double S[9], double D[9]) {
    const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (det == 0.) return false;
    const double d = 1. / det;
    const double t[9] = {(S[4] * S[8] - S[5] * S[7]) * d, (S[2] * S[7] - S[1] * S[8]) * d, (S[1] * S[5] - S[2] * S[4]) * d,
                         (S[5] * S[6] - S[3] * S[8]) * d, (S[0] * S[8] - S[2] * S[6]) * d, (S[2] * S[3] - S[0] * S[5]) * d,
                         (S[3] * S[7] - S[4] * S[6]) * d, (S[1] * S[6] - S[0] * S[7]) * d, (S[0] * S[4] - S[1] * S[3]) * d};
    for (int i = 0; i < 9; i++) D[i] = t[i];
    return true;
}

void launch_warp_perspective(hipStream_t st, const uint8_t *src, int sw, int sh, const double Minv[9], int dw, int dh, uint8_t *dst) {
    Mat3 M;
    for (int i = 0; i < 9; i++) M.m[i] = Minv[i];
    const int bh0 = dh < 16 ? dh : 16;  // OpenCV's block: 16 rows by 1024/16 columns
    const int bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;
    hipLaunchKernelGGL(k_warp_perspective, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, st, src, sw, sh, M, dw, dh, bw0, dst);
}

void launch_fill_and_blend(hipStream_t st, uint8_t *canvas, int cw, const uint8_t *img, int iw, int ih) {
    const size_t n = (size_t)iw * ih * 3;
    hipLaunchKernelGGL(k_fill_and_blend, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, canvas, cw, img, iw, ih);
}

void launch_column_grey_sums(hipStream_t st, const uint8_t *canvas, int cw, int ch, int *sums) {
    hipLaunchKernelGGL(k_column_grey_sums, dim3((cw + 255) / 256), dim3(256), 0, st, canvas, cw, ch, sums);
}

}  // namespace sa
