// Scale-space extrema: 3x3x3 scan with wave-ballot compaction (fused with the
// gradient planes the later stages sample), then per candidate sub-pixel
// localisation and orientation assignment.
//
// Replaces scale_space_extrema_parallel::operator() and its helpers
// (sift.cpp:227-448).  Floating-point operations are written in the order of a
// scalar evaluation of those functions and contraction is off for this file,
// so localisation is bit-reproducible against a CPU evaluation; the orientation
// histogram is summed in fixed point, so it does not depend on the order of
// its additions: it is the same run to run and whichever kernel variant the
// launch size selects.
#include "angle_table.h"
#include "atan2_fast.h"
#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Reserve `count(flag)` consecutive slots of a per-frame list for the lanes of
// this wave whose flag is set; returns the lane's slot or -1.
__device__ __forceinline__ int wave_append(bool flag, int32_t *counter) {
    const unsigned long long mask = __ballot(flag);
    if (mask == 0) return -1;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    const int rank = __popcll(mask & ((1ull << lane_id()) - 1ull));
    return flag ? base + rank : -1;
}

// Value of `v` in the lane to the left (from_left) or right of this one; the
// end lanes of the wave receive an unspecified value.
template <bool from_left>
__device__ __forceinline__ double neighbour_lane(double v) {
    constexpr int ctrl = from_left ? 0x138 : 0x130;  // DPP wave_shr:1 / wave_shl:1
    // bound_ctrl: the end lane reads zero, so the destination needs no initial value (one move less per half)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

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

// One pass over the six Gaussian levels of an octave that produces both of their per-pixel consumers:
//
//  * the gradient planes of levels 1..3: per pixel the magnitude sqrt(dx^2 + dy^2) and the direction
//    atan2(dy, dx) in degrees, with dx = L(y, x+1) - L(y, x-1), dy = L(y-1, x) - L(y+1, x), exactly the
//    per-sample quantities of get_keypoint_orientations (sift.cpp:410-415) and of the descriptor loop
//    (sift.cpp:507-510).  Orientation and descriptor windows of neighbouring keypoints overlap heavily, so
//    evaluating the square root and the arctangent once per pixel instead of once per window sample removes
//    most of the transcendental work of those stages; the values are the same;
//  * is_pixel_extremum (sift.cpp:264-281) for the three scanned layers, on difference-of-Gaussians values
//    formed in registers (level j+1 minus level j, sift.cpp:173).
//
// A wave owns 62 columns x 4 rows (its first and last lane only carry the halo columns); a thread loads the
// six rows of its column in all six levels into registers, every load in flight together, and takes its left
// and right neighbours from the adjacent lanes, so there is no LDS staging of pixels and no barrier after the
// table copy.  Tiles cover the whole image (the gradient planes need every pixel); the extremum test is
// confined to the BORDER-inset rectangle the reference scans (sift.cpp:188, :233).
// 8 and 32 rows per workgroup (2 and 8 per thread) are slower: 15.1 and 17.9 ms against 14.9; so is a wave that
// marches down its 62 columns with rows y-1..y+1 in registers and one new row loaded per step (6 loads per pixel
// instead of 9): 15.1-16.4 ms with one to three rows fetched ahead.  The kernel's vector ALUs are busy two thirds
// of its time, and a marching wave has a quarter of the independent arithmetic per step to hide their latency with.
constexpr int EX_TW = 62, EX_TH = 16;

// max / min of finite doubles as the bare instruction.  fmax()/fmin() make the compiler quieten possible
// signalling NaNs first (one extra v_max_f64 per operand that comes from memory or from a lane exchange),
// which nearly doubled the instruction count of the scan; the difference-of-Gaussian values are finite.
__device__ __forceinline__ double max_finite(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_finite(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ void extrema_grad_tile(const Pyramid &p, int oct, const TileId tile,
                                                  Candidate *__restrict__ cand, int cand_cap,
                                                  FrameCounters *__restrict__ counters, const double (*knots)[2]) {
    const int w = p.w[oct], h = p.h[oct];
    const int f = tile.z;
    const size_t plane = (size_t)w * h;
    const double *G = p.gauss[oct] + (size_t)f * IMAGES * plane;
    Grad *GR = p.grad[oct] + (size_t)f * SCALES * plane;
    const int lane = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // a wave is one row group
    constexpr int RY = EX_TH / 4;  // consecutive rows per thread
    const int j = tile.x * EX_TW + lane - 1;  // this lane's column (-1 or w for a halo lane at the image edge)
    const int i0 = tile.y * EX_TH + ty * RY;  // first row of this wave
    if (i0 >= h) return;
    // Row and column indices are clamped into the image instead of branching: a clamped value only ever
    // stands in for a neighbour of a pixel on the image border, whose gradient is not evaluated and
    // which lies outside the scanned rectangle.
    const int gx = min(max(j, 0), w - 1);
    double g[IMAGES][RY + 2];
#pragma unroll
    for (int lvl = 0; lvl < IMAGES; lvl++)
#pragma unroll
        for (int r = 0; r < RY + 2; r++)
            g[lvl][r] = G[(size_t)lvl * plane + (size_t)min(max(i0 - 1 + r, 0), h - 1) * w + gx];
    const bool own_column = lane >= 1 && lane <= EX_TW && j < w;

    // ---- gradient planes of levels 1..SCALES
#pragma unroll
    for (int layer = 1; layer <= SCALES; layer++) {
#pragma unroll
        for (int q = 0; q < RY; q++) {
            const int y = i0 + q;
            const double l = neighbour_lane<true>(g[layer][q + 1]), r = neighbour_lane<false>(g[layer][q + 1]);
            Grad gr = {0.0, 0.0};
            // windows only sample 0 < x < w-1, 0 < y < h-1 (sift.cpp:407-409, :506)
            if (j > 0 && j < w - 1 && y > 0 && y < h - 1) {
                const double dx = r - l;
                const double dy = g[layer][q] - g[layer][q + 2];
                gr.mag = sqrt(dx * dx + dy * dy);
                gr.ori = direction_degrees(dy, dx, knots);
            }
            if (own_column && y < h) GR[(size_t)(layer - 1) * plane + (size_t)y * w + j] = gr;
        }
    }

    // ---- extrema of the difference-of-Gaussians levels: images[oct][lvl] = gauss[lvl + 1] - gauss[lvl] (sift.cpp:173)
    double v[NDOG][RY + 2];
#pragma unroll
    for (int lvl = 0; lvl < NDOG; lvl++)
#pragma unroll
        for (int r = 0; r < RY + 2; r++) v[lvl][r] = g[lvl + 1][r] - g[lvl][r];
    const double threshold = floor(0.5 * CONTRAST_THRESHOLD / SCALES * 255);  // sift.cpp:266
    const bool column_tested = own_column && j >= BORDER && j < w - BORDER;
    // A positive centre must be >= its 26 neighbours, a negative one <= them
    // (sift.cpp:275-280), i.e. equal to the maximum (minimum) of the 3x3x3 block.
    // The block extremes are built separably: per level the 3-wide row
    // maximum/minimum of six rows, then the 3-tall column one for this thread's
    // four consecutive pixels; the level loop keeps the last three levels.
    double vmax[3][RY], vmin[3][RY];
#pragma unroll
    for (int lvl = 0; lvl < NDOG; lvl++) {
        double hmax[RY + 2], hmin[RY + 2];
#pragma unroll
        for (int r = 0; r < RY + 2; r++) {
            const double b = v[lvl][r];
            const double a = neighbour_lane<true>(b), c = neighbour_lane<false>(b);
            hmax[r] = max_finite(max_finite(a, b), c);
            hmin[r] = min_finite(min_finite(a, b), c);
        }
#pragma unroll
        for (int q = 0; q < RY; q++) {
            vmax[lvl % 3][q] = max_finite(max_finite(hmax[q], hmax[q + 1]), hmax[q + 2]);
            vmin[lvl % 3][q] = min_finite(min_finite(hmin[q], hmin[q + 1]), hmin[q + 2]);
        }
        if (lvl >= 2) {
            const int img = lvl - 1;  // levels img-1, img, img+1 are in the window
#pragma unroll
            for (int q = 0; q < RY; q++) {
                const int i = i0 + q;
                const double c = v[img][q + 1];
                const double hi = max_finite(max_finite(vmax[0][q], vmax[1][q]), vmax[2][q]);
                const double lo = min_finite(min_finite(vmin[0][q], vmin[1][q]), vmin[2][q]);
                const bool is_ext = column_tested && i >= BORDER && i < h - BORDER && fabs(c) > threshold &&
                                    (c > 0 ? c >= hi : c <= lo);  // sift.cpp:268
                const int slot = wave_append(is_ext, &counters[f].candidates);
                if (slot >= 0) {
                    if (slot < cand_cap) {
                        Candidate cd = {oct, img, i, j};
                        cand[(size_t)f * cand_cap + slot] = cd;
                    } else {
                        counters[f].overflow = 1;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_extrema_grad(Pyramid p, int oct, Candidate *__restrict__ cand, int cand_cap,
                                                       FrameCounters *__restrict__ counters) {
    __shared__ double knots[33][2];
    if (threadIdx.x < 66) (&knots[0][0])[threadIdx.x] = (&kAtanKnots[0][0])[threadIdx.x];
    __syncthreads();
    extrema_grad_tile(p, oct, xcd_tile_id(), cand, cand_cap, counters, knots);
}

// The same for several small octaves in one launch (a single frame's upper octaves are one or two tiles each):
// blockIdx.x runs through the tiles of octaves oct_first.. in turn, blockIdx.y is the frame.
__global__ __launch_bounds__(256) void k_extrema_grad_octaves(Pyramid p, int oct_first, Candidate *__restrict__ cand,
                                                               int cand_cap, FrameCounters *__restrict__ counters) {
    __shared__ double knots[33][2];
    if (threadIdx.x < 66) (&knots[0][0])[threadIdx.x] = (&kAtanKnots[0][0])[threadIdx.x];
    __syncthreads();
    int t = blockIdx.x;
    for (int oct = oct_first; oct < p.octaves; oct++) {
        const int w = p.w[oct], h = p.h[oct];
        if (w <= 2 * BORDER || h <= 2 * BORDER || !p.grad[oct]) continue;  // nothing is scanned here (sift.cpp:188, :233)
        const int nx = (w + EX_TW - 1) / EX_TW, ny = (h + EX_TH - 1) / EX_TH;
        if (t < nx * ny) {
            TileId tile;
            tile.x = t % nx;
            tile.y = t / nx;
            tile.z = blockIdx.y;
            extrema_grad_tile(p, oct, tile, cand, cand_cap, counters, knots);
            return;
        }
        t -= nx * ny;
    }
}

// hal::LU64f (eps = DBL_EPSILON*100): in-place LU with partial pivoting, one rhs.
__device__ __forceinline__ bool lu_solve3(double A[3][3], double b[3]) {
    const double eps = 2.220446049250313e-16 * 100;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int k = i;
        double best = fabs(A[i][i]);
#pragma unroll
        for (int j = i + 1; j < 3; j++) {
            const double a = fabs(A[j][i]);
            if (a > best) {
                best = a;
                k = j;
            }
        }
        if (best < eps) return false;
        if (k != i) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j >= i) {
                    // k is 1 or 2 here; select without dynamic register indexing
                    double ai = A[i][j];
                    double ak = (k == 1) ? A[1][j] : A[2][j];
                    A[i][j] = ak;
                    if (k == 1) A[1][j] = ai; else A[2][j] = ai;
                }
            }
            double bi = b[i];
            double bk = (k == 1) ? b[1] : b[2];
            b[i] = bk;
            if (k == 1) b[1] = bi; else b[2] = bi;
        }
        const double d = -1 / A[i][i];
#pragma unroll
        for (int j = i + 1; j < 3; j++) {
            const double alpha = A[j][i] * d;
#pragma unroll
            for (int kk = i + 1; kk < 3; kk++) A[j][kk] += alpha * A[i][kk];
            b[j] += alpha * b[i];
        }
    }
#pragma unroll
    for (int i = 2; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < 3; k++) s -= A[i][k] * b[k];
        b[i] = s / A[i][i];
    }
    return true;
}

// cv::solve(H, g, x, DECOMP_NORMAL) (sift.cpp:332): N = H^T H, r = H^T g, LU.
__device__ __forceinline__ bool solve_normal3(const double H[3][3], const double g[3], double x[3]) {
    double N[3][3], r[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += H[k][i] * H[k][j];
            N[i][j] = s;
            N[j][i] = s;
        }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += H[k][i] * g[k];
        r[i] = s;
    }
    if (!lu_solve3(N, r)) return false;
    x[0] = r[0];
    x[1] = r[1];
    x[2] = r[2];
    return true;
}

// localize_extrema (sift.cpp:312-387).  Returns the layer, 0 if the solve
// failed, -1 if rejected.
__device__ int localize(const Pyramid &p, int f, int oct, int img, int i, int j, DevKeypoint &kpt) {
    const int w = p.w[oct], h = p.h[oct];
    const size_t plane = (size_t)w * h;
    const double *G = p.gauss[oct] + (size_t)f * IMAGES * plane;
    double cube[3][3][3], grad[3], hess[3][3], res[3];
    int attempt;
    for (attempt = 0; attempt < 5; attempt++) {
        const double inv = 1. / 255.0;  // Mat /= 255.0 multiplies by the reciprocal
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++)
                {
                    // DoG level img-1+k = Gaussian level img+k minus Gaussian level img-1+k (sift.cpp:173)
                    const size_t at = (size_t)(img - 1 + k) * plane + (size_t)(i - 1 + a) * w + (j - 1 + b);
                    cube[k][a][b] = (G[at + plane] - G[at]) * inv;
                }
        grad[0] = 0.5 * (cube[1][1][2] - cube[1][1][0]);
        grad[1] = 0.5 * (cube[1][2][1] - cube[1][0][1]);
        grad[2] = 0.5 * (cube[2][1][1] - cube[0][1][1]);
        hess[0][0] = cube[1][1][2] - 2 * cube[1][1][1] + cube[1][1][0];
        hess[1][1] = cube[1][2][1] - 2 * cube[1][1][1] + cube[1][0][1];
        hess[2][2] = cube[2][1][1] - 2 * cube[1][1][1] + cube[0][1][1];
        hess[0][1] = hess[1][0] = 0.25 * (cube[1][2][2] - cube[1][2][0] - cube[1][0][2] + cube[1][0][0]);
        hess[0][2] = hess[2][0] = 0.25 * (cube[2][1][2] - cube[2][1][0] - cube[0][1][2] + cube[0][1][0]);
        hess[1][2] = hess[2][1] = 0.25 * (cube[2][2][1] - cube[2][0][1] - cube[0][2][1] + cube[0][0][1]);
        if (!solve_normal3(hess, grad, res)) return 0;  // sift.cpp:333-335
        res[0] *= -1;
        res[1] *= -1;
        res[2] *= -1;
        if (fabs(res[0]) < 0.5 && fabs(res[1]) < 0.5 && fabs(res[2]) < 0.5) break;  // sift.cpp:339
        j += (int)round(res[0]);
        i += (int)round(res[1]);
        img += (int)round(res[2]);
        if (i < BORDER || i >= h - BORDER || j < BORDER || j >= w - BORDER || img < 1 || img > SCALES) return -1;
    }
    if (attempt == 5) return -1;  // sift.cpp:356
    double dot = 0;
    dot += grad[0] * res[0];
    dot += grad[1] * res[1];
    dot += grad[2] * res[2];
    const double value = cube[1][1][1] + 0.5 * dot;                  // sift.cpp:360
    if (!(fabs(value) * SCALES >= CONTRAST_THRESHOLD)) return -1;    // sift.cpp:363
    const double hess_trace = hess[0][0] + hess[1][1];
    const double hess_det = hess[0][0] * hess[1][1] - hess[0][1] * hess[1][0];
    if (hess_det <= 0) return -1;                                     // sift.cpp:367
    const double r1 = EIGEN_VALUE_RATIO + 1;
    if (!((hess_trace * hess_trace) * EIGEN_VALUE_RATIO < (r1 * r1) * hess_det)) return -1;  // sift.cpp:374
    const int keypt_octave = oct + (1 << 8) * img + (1 << 16) * (int)round((res[2] + 0.5) * 255);  // sift.cpp:375
    const double keypt_pt_x = (j + res[0]) * (1 << oct);
    const double keypt_pt_y = (i + res[1]) * (1 << oct);
    const double keypt_size = SIGMA * exp2(img + res[2] / SCALES) * (1 << (oct + 1));  // sift.cpp:378
    kpt.x = (float)keypt_pt_x;
    kpt.y = (float)keypt_pt_y;
    kpt.size = (float)keypt_size;
    kpt.angle = -1.0f;
    kpt.response = (float)fabs(value);
    kpt.octave = keypt_octave;
    kpt.class_id = -1;
    kpt.pad = 0;
    return img;
}

// localize_extrema (sift.cpp:312-387), one thread per candidate.  Survivors are
// appended to a per-frame list as keypoints with angle -1; the layer the
// localisation ended on (where the orientation is measured) rides in `pad`.
__global__ __launch_bounds__(256) void k_localize(Pyramid p, const Candidate *__restrict__ cand, int cand_cap,
                                                   DevKeypoint *__restrict__ localized, int kp_cap,
                                                   FrameCounters *__restrict__ counters,
                                                   int32_t *__restrict__ n_localized) {
    const int f = blockIdx.y;
    int n = counters[f].candidates;
    if (n > cand_cap) n = cand_cap;
    if ((int)(blockIdx.x * 256) >= n) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    DevKeypoint kpt;
    int layer = -1;
    int oct = 0;
    if (idx < n) {
        const Candidate cd = cand[(size_t)f * cand_cap + idx];
        oct = cd.oct;
        layer = localize(p, f, cd.oct, cd.img, cd.i, cd.j, kpt);
    }
    // layer 0 (singular system) runs the orientation of a default keypoint in
    // the reference, which emits nothing (sift.cpp:333-335, :401-409)
    const bool keep = layer > 0;
    const int slot = wave_append(keep, &n_localized[f]);
    if (slot >= 0) {
        if (slot < kp_cap) {
            kpt.pad = layer | (oct << 8);
            localized[(size_t)f * kp_cap + slot] = kpt;
        } else {
            counters[f].overflow = 1;
        }
    }
}

// get_keypoint_orientations (sift.cpp:393-448), one wave per localised
// keypoint; keypoints are dealt round-robin to the waves of their frame.
//
// The 36-bin histogram is kept as 16 partial histograms in LDS, one per
// lane & 15, updated with LDS double atomics; a partial histogram is only ever
// touched by the four lanes of one wave that share it, in program order, so
// the summation order is fixed and results are reproducible.  The Gaussian
// weight exp(wf*(i^2+j^2)) is the product of two entries of a per-keypoint
// table exp(wf*t^2).
constexpr int OCOPIES = 16;
// Word of (bin, copy) = bin * 16 + copy, copy = lane & 15: the LDS takes a 64-bit atomic in four passes of 16
// consecutive lanes, whose copies all differ, so no two lanes of a pass share an address, and their banks
// 2 * copy + 32 * (bin & 1) all differ too, whatever bins they hit.
constexpr int OSTRIDE = OCOPIES;
constexpr int OWAVES = 4;             // waves (= keypoints in flight) per workgroup
constexpr int WTAB = 128;
constexpr int OSAMP = 4;             // samples per lane per trip of the window loop (2 and 8 measure the same)

// TEAM = false: a wave per keypoint (launches with keypoints enough to keep every wave busy).  TEAM = true: the four
// waves of a workgroup share a keypoint, each taking every fourth trip through the window with partial histograms
// of its own, which wave 0 then adds up wave by wave; for one or a few frames, where the stage lasts as long as its
// largest windows.  The histogram is summed in fixed point (64-bit integer LDS atomics on contributions rounded to
// multiples of 2^-k, k = 40 and more), so the two modes, which add the same terms in different orders, give the same
// bins bit for bit, and so does every run.  The conversion is free: the two Gaussian weight factors carry 2^((k-1074)/2)
// each, which puts magnitude * w_row * w_col at n * 2^-1074, where the bit pattern of a double is the integer n.
template <bool TEAM>
__global__ __launch_bounds__(64 * OWAVES) void k_orient(Pyramid p, const DevKeypoint *__restrict__ localized,
                                                         int kp_cap, DevKeypoint *__restrict__ oriented,
                                                         FrameCounters *__restrict__ counters,
                                                         const int32_t *__restrict__ n_localized) {
    __shared__ unsigned long long s_hist[OWAVES][BINS * OSTRIDE];
    __shared__ double s_hsum[OWAVES][BINS];
    __shared__ double s_smooth[OWAVES][BINS];
    __shared__ double s_wtab[OWAVES][WTAB];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long *hist = s_hist[wv];
    double *hsum = s_hsum[TEAM ? 0 : wv], *smooth = s_smooth[TEAM ? 0 : wv], *wtab = s_wtab[TEAM ? 0 : wv];
    int n = n_localized[f];
    if (n > kp_cap) n = kp_cap;
    // keypoints are dealt to the waves of the frame round-robin: a shared cursor would cost one
    // same-address global atomic per keypoint, which serialises the whole frame
    const int first = TEAM ? blockIdx.x : blockIdx.x * OWAVES + wv, stride = TEAM ? gridDim.x : gridDim.x * OWAVES;
    // wave-level hand-over between the lanes of a wave; in TEAM mode the waves meet at workgroup barriers instead
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (int idx = first; idx < n; idx += stride) {
        const DevKeypoint kpt = localized[(size_t)f * kp_cap + idx];
        const int layer = kpt.pad & 255, oct = kpt.pad >> 8;
        const int w = p.w[oct], h = p.h[oct];
        const Grad *L = p.grad[oct] + grad_off(p, oct, f, layer);
        const int base_x = (int)round(kpt.x / double(1 << oct));  // sift.cpp:398
        const int base_y = (int)round(kpt.y / double(1 << oct));
        const double scale = SCALE_FACTOR * kpt.size / double(1 << (oct + 1));  // sift.cpp:401
        const int radius = (int)round(scale * RADIUS_FACTOR);
        const double weight_factor = -0.5 / (scale * scale);
        // fewer than 2^lg samples of less than 2^9 each (a gradient magnitude is at most 255 * sqrt(2)): no bin reaches
        // 2^63; k even, so that each weight factor carries half of the scale
        const unsigned long long area = (unsigned long long)(2 * (long long)radius + 1) * (unsigned long long)(2 * (long long)radius + 1);
        const int fx_bits = min(42, 54 - (64 - __builtin_clzll(area))) & ~1;
        const double half_scale = __longlong_as_double((long long)(1023 + (fx_bits - 1074) / 2) << 52);
        const double unscale = __longlong_as_double((long long)(1023 - fx_bits) << 52);
        for (int t = lane; t < BINS * OSTRIDE; t += 64) hist[t] = 0;
        const bool use_tab = radius < WTAB;
        if (use_tab) {
            if constexpr (TEAM) {
                for (int t = threadIdx.x; t <= radius; t += 64 * OWAVES) wtab[t] = exp(weight_factor * (double)(t * t)) * half_scale;
            } else {
                for (int t = lane; t <= radius; t += 64) wtab[t] = exp(weight_factor * (double)(t * t)) * half_scale;
            }
        }
        if constexpr (TEAM) __syncthreads(); else wave_sync();
        const int n1 = 2 * radius + 1;
        const int total = n1 * n1;
        const int copy = lane & (OCOPIES - 1);
        // The (2 radius + 1)^2 window is walked in row-major order, sample s = lane, lane + 64, ...
        // (row, column) of a lane's sample advance by 64 samples at a time: no division in the loop.
        const int q64 = 64 / n1, r64 = 64 - q64 * n1;
        int qi = lane / n1, rj = lane - qi * n1;
        // Four samples per lane per trip.  Gradient loads are unconditional (addresses clamped
        // into the plane) and the weight-table reads come before the first atomic, so all of a
        // trip's reads are in flight together; samples outside the image are discarded.
        constexpr int TRIP = 64 * OSAMP;
        for (int s0 = lane + (TEAM ? wv * TRIP : 0); s0 < total; s0 += TEAM ? TRIP * OWAVES : TRIP) {
            if constexpr (TEAM) {  // this wave skips the other waves' trips
                qi = s0 / n1;
                rj = s0 - qi * n1;
            }
            Grad g[OSAMP];
            double wrow[OSAMP], wcol[OSAMP];  // exp(-i^2 / 2 s^2) and exp(-j^2 / 2 s^2), each times 2^((k - 1074) / 2)
            bool ok[OSAMP];
#pragma unroll
            for (int u = 0; u < OSAMP; u++) {
                const int ii = qi - radius, jj = rj - radius;
                const int y = base_y + ii, x = base_x + jj;
                ok[u] = s0 + 64 * u < total && y > 0 && y < h - 1 && x > 0 && x < w - 1;  // sift.cpp:407,409
                const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
                g[u] = L[(size_t)yc * w + xc];  // mag, direction (sift.cpp:410-415)
                if (use_tab) {
                    wrow[u] = wtab[min(abs(ii), WTAB - 1)];
                    wcol[u] = wtab[min(abs(jj), WTAB - 1)];
                } else {
                    wrow[u] = exp(weight_factor * (ii * ii + jj * jj)) * half_scale;
                    wcol[u] = half_scale;
                }
                rj += r64;
                qi += q64;
                if (rj >= n1) {
                    rj -= n1;
                    qi++;
                }
            }
#pragma unroll
            for (int u = 0; u < OSAMP; u++) {
                if (!ok[u]) continue;
                // ((int)round(ori*36/360) % 36 + 36) % 36 (sift.cpp:416); |ori| <= 180, so the rounded value lies in [-18, 18]
                const int k = round_half_away(div_360(g[u].ori * BINS));
                const int index = k < 0 ? k + BINS : k;
                // sift.cpp:417: the first product is an ordinary number, the second lands on the integer grid
                atomicAdd(&hist[index * OSTRIDE + copy], (unsigned long long)__double_as_longlong((g[u].mag * wrow[u]) * wcol[u]));
            }
        }
        if constexpr (TEAM) __syncthreads(); else wave_sync();
        if (TEAM && wv != 0) {
            __syncthreads();  // the one at the end of the trip: wave 0 is done with the histograms and the table
            continue;
        }
        if (lane < BINS) {
            unsigned long long t = 0;
#pragma unroll
            for (int v = 0; v < (TEAM ? OWAVES : 1); v++) {
                const unsigned long long *hv = TEAM ? s_hist[v] : hist;
#pragma unroll
                for (int c = 0; c < OCOPIES; c++) t += hv[lane * OSTRIDE + c];
            }
            hsum[lane] = (double)t * unscale;
        }
        wave_sync();
        if (lane < BINS) {
            const int i = lane;
            smooth[i] = (6 * hsum[i] + 4 * (hsum[(i + BINS - 1) % BINS] + hsum[(i + 1) % BINS]) +
                         hsum[(i + BINS - 2) % BINS] + hsum[(i + 2) % BINS]) / 16.;  // sift.cpp:426-429
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double max_orientation = smooth[0];
        for (int b = 1; b < BINS; b++) max_orientation = fmax(max_orientation, smooth[b]);
        bool is_peak = false;
        double orientation = 0;
        if (lane < BINS) {
            const int i = lane;
            const double l = smooth[(i + BINS - 1) % BINS], r = smooth[(i + 1) % BINS];
            const double peak = smooth[i];
            if (peak > l && peak > r && peak >= PEAK_RATIO * max_orientation) {  // sift.cpp:435-437
                const double interpolated_index = fmod((i + 0.5 * (l - r) / (l + r - 2 * peak)), (double)BINS);
                orientation = 360 - interpolated_index * 360 / BINS;
                if (fabs(360 - orientation) < EPS) orientation = 0;
                is_peak = true;
            }
        }
        const int slot = wave_append(is_peak, &counters[f].oriented);
        if (slot >= 0) {
            if (slot < kp_cap) {
                DevKeypoint nk = kpt;
                nk.angle = (float)orientation;
                nk.pad = 0;
                oriented[(size_t)f * kp_cap + slot] = nk;
            } else {
                counters[f].overflow = 1;
            }
        }
        if constexpr (TEAM) __syncthreads(); else wave_sync();
    }
}

}  // namespace

void launch_extrema_gradients(hipStream_t st, const Pyramid &p, int oct, int nframes, Candidate *cand, int cand_cap,
                              FrameCounters *counters) {
    const int w = p.w[oct], h = p.h[oct];
    // empty scan range (sift.cpp:188, :233): no keypoint can come from this octave, so nothing samples its gradients
    if (w <= 2 * BORDER || h <= 2 * BORDER || !p.grad[oct]) return;
    dim3 grid((w + EX_TW - 1) / EX_TW, (h + EX_TH - 1) / EX_TH, nframes);
    hipLaunchKernelGGL(k_extrema_grad, grid, dim3(256), 0, st, p, oct, cand, cand_cap, counters);
}

void launch_extrema_gradients_from(hipStream_t st, const Pyramid &p, int oct_first, int nframes, Candidate *cand,
                                   int cand_cap, FrameCounters *counters) {
    int tiles = 0;
    for (int oct = oct_first; oct < p.octaves; oct++) {
        const int w = p.w[oct], h = p.h[oct];
        if (w <= 2 * BORDER || h <= 2 * BORDER || !p.grad[oct]) continue;
        tiles += ((w + EX_TW - 1) / EX_TW) * ((h + EX_TH - 1) / EX_TH);
    }
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_extrema_grad_octaves, dim3(tiles, nframes), dim3(256), 0, st, p, oct_first, cand, cand_cap, counters);
}

void launch_localize_orient(hipStream_t st, const Pyramid &p, int nframes, const Candidate *cand, int cand_cap,
                            DevKeypoint *localized, DevKeypoint *oriented, int kp_cap, FrameCounters *counters,
                            int32_t *n_localized, bool few_keypoints) {
    hipLaunchKernelGGL(k_localize, dim3((cand_cap + 255) / 256, nframes), dim3(256), 0, st, p, cand, cand_cap,
                       localized, kp_cap, counters, n_localized);
    if (few_keypoints)
        hipLaunchKernelGGL(k_orient<true>, dim3(1024, nframes), dim3(64 * OWAVES), 0, st, p, localized, kp_cap, oriented,
                           counters, n_localized);
    else
        hipLaunchKernelGGL(k_orient<false>, dim3(1024 / OWAVES, nframes), dim3(64 * OWAVES), 0, st, p, localized, kp_cap,
                           oriented, counters, n_localized);
}

}  // namespace sa
