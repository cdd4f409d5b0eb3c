// Gradient planes of Gaussian levels 1..3: per pixel the magnitude
// sqrt(dx^2 + dy^2) and the direction atan2(dy, dx) in degrees, with
// dx = L(y, x+1) - L(y, x-1), dy = L(y-1, x) - L(y+1, x), exactly the per-sample
// quantities of get_keypoint_orientations (sift.cpp:410-415) and of the
// descriptor loop (sift.cpp:507-510).  Orientation and descriptor windows of
// neighbouring keypoints overlap heavily, so evaluating the square root and the
// arctangent once per pixel instead of once per window sample removes most of
// the transcendental work of those stages; the values are the same.
#include "angle_table.h"
#include "atan2_fast.h"
#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

__device__ const double kAtanKnots[33][2] = SA_ATAN_KNOTS;

// Direction in degrees, (atan2(dy,dx)*180)/PI, as host libm would produce it.
// The orientation histogram rounds direction*36/360 to the nearest bin, which
// is discontinuous at the 36 boundaries (5 + 10 m degrees), and images with
// exact diagonal symmetry put strong gradients within an ulp of them, where
// the last bit of atan2 picks the bin.  Host libm's atan2 is correctly rounded
// there; atan2_fast is good to an ulp.  So when the fast result lands within
// 1e-9 bins of a boundary, the angle is recomputed as boundary +
// atan(cross/dot) against a double-double table of the boundary directions,
// which rounds correctly.  `knots` is the LDS copy of kAtanKnots.
__device__ __forceinline__ double direction_degrees(double dy, double dx, const double (*knots)[2]) {
    double a = atan2_fast(dy, dx, knots);
    double orientation = div_pi(a * 180);
    const double v = orientation * (BINS / 360.0);
    const double fl = floor(v);
    if (fabs(v - fl - 0.5) < 1e-9) {
        const int m = (int)fl;
        if (m >= -18 && m <= 17) {
            const double *T = kBinBoundary[m + 18];
            const double p1 = dy * T[2], e1 = fma(dy, T[2], -p1);  // dy*cos(b), exact as p1 + e1
            const double p2 = dx * T[4], e2 = fma(dx, T[4], -p2);  // dx*sin(b)
            // p1 and p2 agree to ~1e-10 here, so p1 - p2 is exact
            const double cross = (p1 - p2) + ((e1 - e2) + (dy * T[3] - dx * T[5]));
            const double dot = dx * T[2] + dy * T[4];
            const double delta = cross / dot;  // |delta| < 1e-9: atan(delta) == delta to 1e-27
            a = T[0] + (T[1] + delta);
            orientation = div_pi(a * 180);
        }
    }
    return orientation;
}

// grid: (ceil(w/64), ceil(h/32), frames*SCALES), renumbered per XCD.  A wave
// owns 64 columns x 8 rows: it loads the ten centre-column rows and the left and
// right neighbours of its eight rows into registers first (every load in
// flight together), then evaluates the eight pixels.
constexpr int GR_ROWS = 8;             // rows per wave
constexpr int GR_TH = 4 * GR_ROWS;     // rows per workgroup

__global__ __launch_bounds__(256) void k_gradients(const double *__restrict__ gauss, Grad *__restrict__ grad, int w,
                                                    int h) {
    __shared__ double knots[33][2];
    if (threadIdx.x < 66) (&knots[0][0])[threadIdx.x] = (&kAtanKnots[0][0])[threadIdx.x];
    __syncthreads();
    const TileId tile = xcd_tile_id();
    const int x = tile.x * 64 + (threadIdx.x & 63);
    const int y0 = tile.y * GR_TH + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * GR_ROWS;  // a wave is one row group
    if (y0 >= h) return;
    const int frame = tile.z / SCALES, layer = tile.z % SCALES + 1;
    const size_t plane = (size_t)w * h;
    const double *L = gauss + ((size_t)frame * IMAGES + layer) * plane;
    Grad *G = grad + ((size_t)frame * SCALES + (layer - 1)) * plane;
    const int xc = min(x, w - 1), xl = max(xc - 1, 0), xr = min(xc + 1, w - 1);
    double c[GR_ROWS + 2], l[GR_ROWS], r[GR_ROWS];
#pragma unroll
    for (int k = 0; k < GR_ROWS + 2; k++) {
        const int yy = min(max(y0 - 1 + k, 0), h - 1);
        c[k] = L[(size_t)yy * w + xc];
    }
#pragma unroll
    for (int k = 0; k < GR_ROWS; k++) {
        const int yy = min(y0 + k, h - 1);
        l[k] = L[(size_t)yy * w + xl];
        r[k] = L[(size_t)yy * w + xr];
    }
    if (x >= w) return;
#pragma unroll
    for (int k = 0; k < GR_ROWS; k++) {
        const int y = y0 + k;
        if (y >= h) break;
        Grad g = {0.0, 0.0};
        // windows only sample 0 < x < w-1, 0 < y < h-1 (sift.cpp:407-409, :506)
        if (x > 0 && x < w - 1 && y > 0 && y < h - 1) {
            const double dx = r[k] - l[k];
            const double dy = c[k] - c[k + 2];
            g.mag = sqrt(dx * dx + dy * dy);
            g.ori = direction_degrees(dy, dx, knots);
        }
        G[(size_t)y * w + x] = g;
    }
}

}  // namespace

void launch_gradients(hipStream_t st, const Pyramid &p, int oct, int nframes) {
    if (!p.grad[oct]) return;
    const int w = p.w[oct], h = p.h[oct];
    dim3 grid((w + 63) / 64, (h + GR_TH - 1) / GR_TH, nframes * SCALES);
    hipLaunchKernelGGL(k_gradients, grid, dim3(256), 0, st, p.gauss[oct], p.grad[oct], w, h);
}

}  // namespace sa
