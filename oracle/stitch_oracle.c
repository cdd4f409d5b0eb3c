/*
 * CPU restatement of the reference's panorama composition, src/main.cpp:115-145:
 * warpPerspective of the running panorama into a canvas as wide as both images,
 * per-channel hole filling from the new image, addWeighted(0.7, 0.3) blend over
 * the new image's rectangle, and the crop by the number of all-black columns.
 * TEST INFRASTRUCTURE ONLY: nothing in the product links or imports this file.
 *
 * PARITY UNPINNED.  warpPerspective, addWeighted, cvtColor and reduce live in
 * OpenCV, which the reference neither vendors nor pins and which is not
 * installed here; the reference holds no stitched output.  Restated from the
 * structure of OpenCV 4.5:
 *   modules/core/src/lapack.cpp         cv::invert, the closed form for 3x3
 *   modules/imgproc/src/imgwarp.cpp     WarpPerspectiveInvoker (64x16 blocks, coordinates in 1/32 pixel),
 *                                       initInterTab2D / remapBilinear (15-bit weights, BORDER_CONSTANT)
 *   modules/core/src/arithm.simd.hpp    addWeighted on 8-bit data: float, saturate_cast (assumed unfused)
 *   modules/imgproc/src/color_rgb.simd.hpp  BGR2GRAY, 15-bit coefficients
 */
#include "stitch_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_invert3(const double S[9], double D[9]) {
    const double d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d0 == 0.) return 0;
    const double d = 1. / d0;
    double t[9];
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    memcpy(D, t, sizeof t);
    return 1;
}

static int sat_int(double v) { /* saturate_cast<int>(double) = cvRound, on a value already clamped to the int range */
    return (int)lrint(v);
}
static short sat_short(int v) { return (short)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v); }

/* The four 15-bit weights of the fractional position (fx, fy) in 1/32 pixel (initInterTab2D, fixed point). */
static void bilinear_weights(int fx, int fy, int w[4]) {
    const float x = fx * (1.f / 32), y = fy * (1.f / 32);
    const float tx[2] = {1.f - x, x}, ty[2] = {1.f - y, y};
    int isum = 0;
    for (int k1 = 0; k1 < 2; k1++)
        for (int k2 = 0; k2 < 2; k2++) {
            const float v = ty[k1] * tx[k2] * 32768;
            long r = lrintf(v);
            if (r > SHRT_MAX) r = SHRT_MAX;
            w[2 * k1 + k2] = (int)r;
            isum += (int)r;
        }
    /* the sum must be 32768; it only is not at (0, 0), where 32768 saturates to 32767 and the table's
       correction lands on the last of the four weights */
    if (isum != 32768) w[3] -= isum - 32768;
}

int orc_warp_perspective(const uint8_t *src, int sw, int sh, const double H[9], int dw, int dh, uint8_t *dst) {
    double M[9];
    if (!orc_invert3(H, M)) return 0;
    const int bh0 = dh < 16 ? dh : 16;                      /* block of BLOCK_SZ/2 rows ... */
    const int bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;     /* ... by BLOCK_SZ*BLOCK_SZ/rows columns: 64 x 16 */
    for (int y = 0; y < dh; y++) {
        for (int xb = 0; xb < dw; xb += bw0) {
            const double X0 = M[0] * xb + M[1] * y + M[2];
            const double Y0 = M[3] * xb + M[4] * y + M[5];
            const double W0 = M[6] * xb + M[7] * y + M[8];
            const int bw = dw - xb < bw0 ? dw - xb : bw0;
            for (int x1 = 0; x1 < bw; x1++) {
                double W = W0 + M[6] * x1;
                W = W ? 32 / W : 0;
                const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + M[0] * x1) * W));
                const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + M[3] * x1) * W));
                const int X = sat_int(fX), Y = sat_int(fY);
                const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
                int w[4];
                bilinear_weights(X & 31, Y & 31, w);
                uint8_t *D = dst + ((size_t)y * dw + xb + x1) * 3;
                if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                    D[0] = D[1] = D[2] = 0;
                    continue;
                }
                for (int k = 0; k < 3; k++) {
                    int v[4];
                    for (int q = 0; q < 4; q++) {
                        const int px = sx + (q & 1), py = sy + (q >> 1);
                        v[q] = (px >= 0 && px < sw && py >= 0 && py < sh) ? src[((size_t)py * sw + px) * 3 + k] : 0;
                    }
                    const int sum = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                    const int r = (sum + (1 << 14)) >> 15;
                    D[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                }
            }
        }
    }
    return 1;
}

static uint8_t add_weighted(uint8_t a, uint8_t b) {
    const float v = (float)a * 0.7f + (float)b * 0.3f + 0.f;
    const long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

int orc_stitch_pair(const uint8_t *ref, int rw, int rh, const uint8_t *img, int iw, int ih, const double H[9],
                    uint8_t *out, int *out_w) {
    const int W = rw + iw;
    if (ih > rh) return 0; /* result(Rect(0, 0, img.cols, img.rows)) throws */
    uint8_t *res = malloc((size_t)W * rh * 3);
    if (!orc_warp_perspective(ref, rw, rh, H, W, rh, res)) {
        free(res);
        return 0;
    }
    for (int i = 0; i < ih; i++)
        for (int j = 0; j < iw; j++)
            for (int l = 0; l < 3; l++) {
                uint8_t *p = res + ((size_t)i * W + j) * 3 + l;
                const uint8_t q = img[((size_t)i * iw + j) * 3 + l];
                if (*p == 0) *p = q;      /* main.cpp:126-128 */
                *p = add_weighted(q, *p); /* main.cpp:133-134 */
            }
    int remove_cols = 0;
    for (int j = 0; j < W; j++) {
        int sum = 0;
        for (int i = 0; i < rh; i++) {
            const uint8_t *p = res + ((size_t)i * W + j) * 3;
            sum += (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
        }
        remove_cols += sum == 0; /* main.cpp:141 */
    }
    const int nw = W - remove_cols;
    for (int i = 0; i < rh; i++) memcpy(out + (size_t)i * nw * 3, res + (size_t)i * W * 3, (size_t)nw * 3);
    *out_w = nw;
    free(res);
    return 1;
}
