// RANSAC homography between matched keypoints: the consumer of the matcher's
// output, cv::findHomography(obj, scene, RANSAC, 3, mask) at main.cpp:105.
//
// OpenCV's loop (calib3d ptsetreg.cpp, fundam.cpp) is sequential: draw four
// correspondences, fit a model, count the inliers, keep the best, shrink the
// iteration budget.  Only the draw depends on the random generator and it does
// not depend on earlier models, so the draws of all iterations are replayed on
// the host first (microseconds), the models and inlier counts of a whole chunk
// of iterations are evaluated on the GPU at once (one thread per 4-point model,
// one wave per model for the count over all correspondences), and the host then
// walks the counts in iteration order, applying the same "better than the best
// so far" and iteration-budget rules.  The outcome equals the sequential loop's.
// The final fit over the inliers and the Levenberg-Marquardt polish are
// sequential sums over a few hundred points and run on the host in the order
// OpenCV uses.
//
// Every double operation below is a separate IEEE operation (contraction is
// off for this library), square roots and divisions are correctly rounded on
// both sides, so host and device produce the same bits.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

#define SA_HOSTDEV __host__ __device__ __forceinline__

// ---- element access ------------------------------------------------------------
template <class T>
struct Plain {
    T *p;
    SA_HOSTDEV T &operator[](int i) const { return p[i]; }
    SA_HOSTDEV Plain offset(int i) const { return Plain{p + i}; }
};
// hypot as cv::hypot computes it (core/base.hpp)
SA_HOSTDEV double scaled_hypot(double a, double b) {
    a = fabs(a);
    b = fabs(b);
    if (a > b) {
        b /= a;
        return a * sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * sqrt(1 + a * a);
    }
    return 0;
}

// Largest off-diagonal magnitude right of the diagonal in row k / above it in column k.
template <class AD>
SA_HOSTDEV int row_pivot(const AD &A, int n, int k) {
    int m = k + 1;
    double mv = fabs(A[n * k + m]);
    for (int i = k + 2; i < n; i++) {
        const double val = fabs(A[n * k + i]);
        if (mv < val) mv = val, m = i;
    }
    return m;
}
template <class AD>
SA_HOSTDEV int col_pivot(const AD &A, int n, int k) {
    int m = 0;
    double mv = fabs(A[k]);
    for (int i = 1; i < k; i++) {
        const double val = fabs(A[n * i + k]);
        if (mv < val) mv = val, m = i;
    }
    return m;
}

// Eigen-decomposition of a symmetric n x n matrix by cyclic-by-pivot Jacobi
// rotations, as cv::eigen does it (core lapack.cpp, JacobiImpl_): eigenvalues
// descending in W, eigenvectors in the rows of V.  A is destroyed.
template <class AD, class AI>
SA_HOSTDEV void jacobi_symmetric(const AD &A, const AD &W, const AD &V, const AI &indR, const AI &indC, int n) {
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) V[i * n + j] = 0;
        V[i * n + i] = 1;
    }
    for (int k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) indR[k] = row_pivot(A, n, k);
        if (k > 0) indC[k] = col_pivot(A, n, k);
    }
    const int max_rotations = n * n * 30;
    for (int it = 0; n > 1 && it < max_rotations; it++) {
        int k = 0;
        double mv = fabs(A[indR[0]]);
        for (int i = 1; i < n - 1; i++) {
            const double val = fabs(A[n * i + indR[i]]);
            if (mv < val) mv = val, k = i;
        }
        int l = indR[k];
        for (int i = 1; i < n; i++) {
            const double val = fabs(A[n * indC[i] + i]);
            if (mv < val) mv = val, k = indC[i], l = i;
        }
        const double p = A[n * k + l];
        if (fabs(p) <= DBL_EPSILON) break;
        const double y = (W[l] - W[k]) * 0.5;
        double t = fabs(y) + scaled_hypot(p, y);
        double s = scaled_hypot(p, t);
        const double c = t / s;
        s = p / s;
        t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] = W[k] - t;
        W[l] = W[l] + t;
        auto rotate = [&](const AD &X, int i0, int i1) {
            const double a0 = X[i0], b0 = X[i1];
            X[i0] = a0 * c - b0 * s;
            X[i1] = a0 * s + b0 * c;
        };
        for (int i = 0; i < k; i++) rotate(A, n * i + k, n * i + l);
        for (int i = k + 1; i < l; i++) rotate(A, n * k + i, n * i + l);
        for (int i = l + 1; i < n; i++) rotate(A, n * k + i, n * l + i);
        for (int i = 0; i < n; i++) rotate(V, n * k + i, n * l + i);
        for (int j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : l;
            if (idx < n - 1) indR[idx] = row_pivot(A, n, idx);
            if (idx > 0) indC[idx] = col_pivot(A, n, idx);
        }
    }
    for (int k = 0; k < n - 1; k++) {
        int m = k;
        for (int i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            const double w = W[m];
            W[m] = W[k];
            W[k] = w;
            for (int i = 0; i < n; i++) {
                const double v = V[n * m + i];
                V[n * m + i] = V[n * k + i];
                V[n * k + i] = v;
            }
        }
    }
}

// Normalised direct linear transform (fundam.cpp, HomographyEstimatorCallback::
// runKernel): centre and scale both point sets, accumulate the 9x9 normal matrix
// of the 2 x 9 constraint rows, take the eigenvector of its smallest eigenvalue,
// undo the normalisation, scale so that H[8] = 1.  `from(i)` / `to(i)` return
// correspondence i.  Returns 0 (H untouched) for a degenerate point set.
template <class AD, class AI, class From, class To>
SA_HOSTDEV int fit_dlt(From from, To to, int count, const AD &LtL, const AD &W, const AD &V, const AI &indR,
                       const AI &indC, double H[9]) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
    for (int i = 0; i < count; i++) {
        const float2 M = from(i), m = to(i);
        cmx += m.x;
        cmy += m.y;
        cMx += M.x;
        cMy += M.y;
    }
    cmx /= count;
    cmy /= count;
    cMx /= count;
    cMy /= count;
    double sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
        const float2 M = from(i), m = to(i);
        smx += fabs(m.x - cmx);
        smy += fabs(m.y - cmy);
        sMx += fabs(M.x - cMx);
        sMy += fabs(M.y - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx;
    smy = count / smy;
    sMx = count / sMx;
    sMy = count / sMy;
    for (int i = 0; i < 81; i++) LtL[i] = 0;
    for (int i = 0; i < count; i++) {
        const float2 M = from(i), m = to(i);
        const double x = (m.x - cmx) * smx, y = (m.y - cmy) * smy;
        const double X = (M.x - cMx) * sMx, Y = (M.y - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++) LtL[9 * j + k] = LtL[9 * j + k] + (Lx[j] * Lx[k] + Ly[j] * Ly[k]);
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[9 * j + k] = LtL[9 * k + j];
    jacobi_symmetric(LtL, W, V, indR, indC, 9);
    // H = inv(T_to) * H0 * T_from with the two normalising transforms written out
    const double inv_to[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double t_from[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double h0[9], tmp[9], out[9];
#pragma unroll
    for (int i = 0; i < 9; i++) h0[i] = V[72 + i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) tmp[3 * i + j] = inv_to[3 * i] * h0[j] + inv_to[3 * i + 1] * h0[3 + j] + inv_to[3 * i + 2] * h0[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) out[3 * i + j] = tmp[3 * i] * t_from[j] + tmp[3 * i + 1] * t_from[3 + j] + tmp[3 * i + 2] * t_from[6 + j];
    const double scale = 1. / out[8];
#pragma unroll
    for (int i = 0; i < 9; i++) H[i] = out[i] * scale;
    return 1;
}

// Squared reprojection error in single precision (fundam.cpp, computeError).
struct ModelF {
    float h[8];
};
SA_HOSTDEV ModelF to_float(const double *H) {
    ModelF f;
#pragma unroll
    for (int i = 0; i < 8; i++) f.h[i] = (float)H[i];
    return f;
}
SA_HOSTDEV float reprojection_error(const ModelF &f, float2 M, float2 m) {
    const float ww = 1.f / (f.h[6] * M.x + f.h[7] * M.y + 1.f);
    const float dx = (f.h[0] * M.x + f.h[1] * M.y + f.h[2]) * ww - m.x;
    const float dy = (f.h[3] * M.x + f.h[4] * M.y + f.h[5]) * ww - m.y;
    return dx * dx + dy * dy;
}

// ---- device side ---------------------------------------------------------------

// A 4-point model per team of 16 lanes, four teams per wave.  The 9x9 eigenproblem is
// the long pole (a hundred-odd dependent Jacobi rotations), so a team splits each rotation:
// lane i rotates the pair of matrix entries and the pair of eigenvector entries of index i,
// the pivot caches (one per lane: the row maxima in lanes 0..7, the column maxima in lanes
// 8..15) are searched and refreshed with 16-lane arg-max reductions that prefer the lowest
// position among equals, which is what the sequential scans with their strict comparisons
// pick.  Every entry is computed from the same operands by the same operations as in
// jacobi_symmetric, so the result has the same bits.
constexpr int TEAM = 16, TEAMS = 4;

// Lane exchange inside a row of 16 lanes with DPP moves (no LDS round trip): the two quad
// permutations make every lane hold its quad's result, the rotations by 4 and 8 lanes then
// combine the four quads.
template <int CTRL>
__device__ __forceinline__ int dpp_int(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ void argmax_step(double &v, int &key) {
    const double ov = __hiloint2double(dpp_int<CTRL>(__double2hiint(v)), dpp_int<CTRL>(__double2loint(v)));
    const int okey = dpp_int<CTRL>(key);
    const bool take = ov > v || (ov == v && okey < key);
    v = take ? ov : v;
    key = take ? okey : key;
}
// All 16 lanes of a team end with the largest v and the key that came with it; among equal
// values the smallest key wins.
__device__ __forceinline__ void team_argmax(double &v, int &key) {
    argmax_step<0xB1>(v, key);   // quad_perm [1,0,3,2]
    argmax_step<0x4E>(v, key);   // quad_perm [2,3,0,1]
    argmax_step<0x124>(v, key);  // row_ror:4
    argmax_step<0x128>(v, key);  // row_ror:8
}

__global__ __launch_bounds__(TEAM *TEAMS) void k_h_models(const float2 *__restrict__ src, const float2 *__restrict__ dst,
                                                          const int *__restrict__ subsets, int nh,
                                                          double *__restrict__ models, uint8_t *__restrict__ valid) {
    __shared__ double sA[TEAMS][81], sV[TEAMS][81], sW[TEAMS][9], sL[TEAMS][4][18];
    __shared__ int sPerm[TEAMS][9];
    const int team = threadIdx.x / TEAM, j = threadIdx.x % TEAM;
    const int h = blockIdx.x * TEAMS + team;
    const bool live = h < nh;
    const int *sub = subsets + 4 * (size_t)(live ? h : nh - 1);  // idle teams shadow the last model and write nothing
    double *A = sA[team], *V = sV[team], *W = sW[team];

    // normalisation, by every lane (same values): fit_dlt with count = 4
    float2 M[4], m[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        M[p] = src[sub[p]];
        m[p] = dst[sub[p]];
    }
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        cmx += m[p].x;
        cmy += m[p].y;
        cMx += M[p].x;
        cMy += M[p].y;
    }
    cmx /= 4;
    cmy /= 4;
    cMx /= 4;
    cMy /= 4;
    double sMx = 0, sMy = 0, smx = 0, smy = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        smx += fabs(m[p].x - cmx);
        smy += fabs(m[p].y - cmy);
        sMx += fabs(M[p].x - cMx);
        sMy += fabs(M[p].y - cMy);
    }
    const bool ok = !(fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON);
    smx = ok ? 4 / smx : 1.0;  // a degenerate subset goes through the motions on finite numbers and is flagged invalid
    smy = ok ? 4 / smy : 1.0;
    sMx = ok ? 4 / sMx : 1.0;
    sMy = ok ? 4 / sMy : 1.0;

    // the two constraint rows of point j (lanes 0..3)
    if (j < 4) {
        const float2 Mj = j == 0 ? M[0] : j == 1 ? M[1] : j == 2 ? M[2] : M[3];
        const float2 mj = j == 0 ? m[0] : j == 1 ? m[1] : j == 2 ? m[2] : m[3];
        const double x = (mj.x - cmx) * smx, y = (mj.y - cmy) * smy;
        const double X = (Mj.x - cMx) * sMx, Y = (Mj.y - cMy) * sMy;
        double *L = sL[team][j];
        L[0] = X, L[1] = Y, L[2] = 1, L[3] = 0, L[4] = 0, L[5] = 0, L[6] = -x * X, L[7] = -x * Y, L[8] = -x;
        L[9] = 0, L[10] = 0, L[11] = 0, L[12] = X, L[13] = Y, L[14] = 1, L[15] = -y * X, L[16] = -y * Y, L[17] = -y;
    }
    __syncthreads();
    // row j of the normal matrix (upper part, mirrored), identity eigenvectors, diagonal
    if (j < 9) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (k >= j) {
                double acc = 0;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const double *L = sL[team][p];
                    acc = acc + (L[j] * L[k] + L[9 + j] * L[9 + k]);
                }
                A[9 * j + k] = acc;
                A[9 * k + j] = acc;
                if (k == j) W[j] = acc;
            }
            V[9 * j + k] = k == j ? 1.0 : 0.0;
        }
    }
    __syncthreads();

    const Plain<double> Am{A};
    int my_ind = j < 8 ? row_pivot(Am, 9, j) : col_pivot(Am, 9, j - 7);  // indR[j] | indC[j-7]
    bool active = true;
    for (int it = 0; it < 9 * 9 * 30; it++) {
        // pivot: first maximum over the row caches 0..7, then the column caches 1..8
        const int ck = j < 8 ? j : my_ind, cl = j < 8 ? my_ind : j - 7;
        double best = fabs(A[9 * ck + cl]);
        int key = j * 256 + ck * 16 + cl;  // position first: the earliest of equal maxima wins
        team_argmax(best, key);
        const int k = (key >> 4) & 15, l = key & 15;
        const double p = A[9 * k + l];
        if (fabs(p) <= DBL_EPSILON) active = false;
        if (!__any(active)) break;
        const double wk = W[k], wl = W[l];
        const double y = (wl - wk) * 0.5;
        double t = fabs(y) + scaled_hypot(p, y);
        double s = scaled_hypot(p, t);
        const double c = t / s;
        s = p / s;
        t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        // lane i < 9 owns the entry pair of index i (none for i = k, l) and the eigenvector pair of column i
        const bool has_v = j < 9, has_a = has_v && j != k && j != l;
        const int jj = has_v ? j : 0;
        const int a0i = jj < k ? 9 * jj + k : 9 * k + jj, a1i = jj < l ? 9 * jj + l : 9 * l + jj;
        const double a0 = A[a0i], a1 = A[a1i], v0 = V[9 * k + jj], v1 = V[9 * l + jj];
        __syncthreads();  // every lane has read what it rotates (and p) before anything is overwritten
        if (active) {
            if (has_a) {
                A[a0i] = a0 * c - a1 * s;
                A[a1i] = a0 * s + a1 * c;
            }
            if (has_v) {
                V[9 * k + jj] = v0 * c - v1 * s;
                V[9 * l + jj] = v0 * s + v1 * c;
            }
            if (j == 0) {
                A[9 * k + l] = 0;
                W[k] = wk - t;
                W[l] = wl + t;
            }
        }
        __syncthreads();
        // refresh the caches of rows and columns k and l
        double rk = j > k && j < 9 ? fabs(A[9 * k + j]) : -1.0, ck2 = j < k ? fabs(A[9 * j + k]) : -1.0;
        double rl = j > l && j < 9 ? fabs(A[9 * l + j]) : -1.0, cl2 = j < l ? fabs(A[9 * j + l]) : -1.0;
        int prk = j, pck = j, prl = j, pcl = j;
        team_argmax(rk, prk);
        team_argmax(ck2, pck);
        team_argmax(rl, prl);
        team_argmax(cl2, pcl);
        if (active) {
            if (j < 8) {
                if (j == k) my_ind = prk;  // k < l <= 8
                if (j == l) my_ind = prl;
            } else {
                const int i = j - 7;
                if (i == k) my_ind = pck;  // k >= 1 here
                if (i == l) my_ind = pcl;
            }
        }
    }
    __syncthreads();
    // which eigenvector does the descending selection sort leave in the last row?
    if (j == 0) {
        int *perm = sPerm[team];
        for (int i = 0; i < 9; i++) perm[i] = i;
        for (int k = 0; k < 8; k++) {
            int mi = k;
            for (int i = k + 1; i < 9; i++)
                if (W[mi] < W[i]) mi = i;
            if (k != mi) {
                const double w = W[mi];
                W[mi] = W[k];
                W[k] = w;
                const int q = perm[mi];
                perm[mi] = perm[k];
                perm[k] = q;
            }
        }
    }
    __syncthreads();
    if (j == 0 && live) {
        const double *h0 = V + 9 * sPerm[team][8];
        const double inv_to[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
        const double t_from[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
        double tmp[9], out[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int q = 0; q < 3; q++) tmp[3 * r + q] = inv_to[3 * r] * h0[q] + inv_to[3 * r + 1] * h0[3 + q] + inv_to[3 * r + 2] * h0[6 + q];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int q = 0; q < 3; q++) out[3 * r + q] = tmp[3 * r] * t_from[q] + tmp[3 * r + 1] * t_from[3 + q] + tmp[3 * r + 2] * t_from[6 + q];
        const double scale = 1. / out[8];
        valid[h] = (uint8_t)ok;
        if (ok) {
#pragma unroll
            for (int i = 0; i < 9; i++) models[9 * (size_t)h + i] = out[i] * scale;
        }
    }
}

// One wave per model: counts[h] = number of correspondences within the threshold.
constexpr int SCORE_WAVES = 4;
__global__ __launch_bounds__(64 * SCORE_WAVES) void k_h_score(const float2 *__restrict__ src, const float2 *__restrict__ dst,
                                                              int n, const double *__restrict__ models,
                                                              const uint8_t *__restrict__ valid, int nh, float thresh2,
                                                              int *__restrict__ counts) {
    const int h = blockIdx.x * SCORE_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (h >= nh) return;
    int cnt = 0;
    if (valid[h]) {
        const ModelF f = to_float(models + 9 * (size_t)h);
        for (int i = lane; i < n; i += 64) cnt += reprojection_error(f, src[i], dst[i]) <= thresh2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) counts[h] = cnt;
}

__global__ __launch_bounds__(256) void k_h_mask(const float2 *__restrict__ src, const float2 *__restrict__ dst, int n, ModelF f,
                                                float thresh2, uint8_t *__restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) mask[i] = reprojection_error(f, src[i], dst[i]) <= thresh2;
}

// ---- host side -----------------------------------------------------------------

// cv::RNG: multiply-with-carry (core/operations.hpp)
struct Mwc {
    uint64_t state;
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// Is the last of `count` points on a line through two earlier ones, or too close to them
// (fundam.cpp, haveCollinearPoints)?
bool last_point_collinear(const float2 *p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[j].x - p[i].x, dy1 = p[j].y - p[i].y;
        for (int k = 0; k < j; k++) {
            const double dx2 = p[k].x - p[i].x, dy2 = p[k].y - p[i].y;
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

double triangle_det(const float2 &a, const float2 &b, const float2 &c) {
    // determinant of [a 1; b 1; c 1] expanded along the first row (Matx33d)
    const double a00 = a.x, a01 = a.y, a10 = b.x, a11 = b.y, a20 = c.x, a21 = c.y;
    return a00 * (a11 * 1. - a21 * 1.) - a01 * (a10 * 1. - a20 * 1.) + 1. * (a10 * a21 - a20 * a11);
}

// HomographyEstimatorCallback::checkSubset: no collinear triple with the last point, and
// the four triangles keep their orientation under the mapping (all or none flipped).
bool subset_usable(const float2 *s, const float2 *d, int count) {
    if (last_point_collinear(s, count) || last_point_collinear(d, count)) return false;
    if (count == 4) {
        static const int tri[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (const auto &t : tri)
            negative += triangle_det(s[t[0]], s[t[1]], s[t[2]]) * triangle_det(d[t[0]], d[t[1]], d[t[2]]) < 0;
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}

// RANSACPointSetRegistrator::getSubset (ptsetreg.cpp): four distinct indices, redrawn as a
// whole (up to max_attempts times) while the subset is unusable.
bool draw_subset(const float2 *src, const float2 *dst, int count, Mwc &rng, int max_attempts, int idx[4]) {
    float2 s[4], d[4];
    for (int attempt = 0; attempt < max_attempts; ++attempt) {
        for (int i = 0; i < 4; ++i) {
            int pick;
            bool repeated;
            do {
                pick = rng.uniform(0, count);
                repeated = false;
                for (int j = 0; j < i; j++) repeated |= idx[j] == pick;
            } while (repeated);
            idx[i] = pick;
            s[i] = src[pick];
            d[i] = dst[pick];
        }
        if (subset_usable(s, d, 4)) return true;
    }
    return false;
}

// RANSACUpdateNumIters (ptsetreg.cpp)
int updated_iteration_budget(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

struct HostEigen {
    double A[81], W[9], V[81];
    int indR[9], indC[9];
    void run(int n) { jacobi_symmetric(Plain<double>{A}, Plain<double>{W}, Plain<double>{V}, Plain<int>{indR}, Plain<int>{indC}, n); }
};

int fit_dlt_host(const float2 *src, const float2 *dst, int count, double H[9]) {
    HostEigen e;
    return fit_dlt([&](int i) { return src[i]; }, [&](int i) { return dst[i]; }, count, Plain<double>{e.A}, Plain<double>{e.W},
                   Plain<double>{e.V}, Plain<int>{e.indR}, Plain<int>{e.indC}, H);
}

// Pseudo-inverse application through the eigen-decomposition of a symmetric 8x8 matrix, as
// cv::solve / cv::invert do with DECOMP_EIG (lapack.cpp: Jacobi, then SVBkSb with the
// threshold 2*eps*sum(w)).  rhs == nullptr produces the inverse's diagonal in x.
void eig_apply(const double S[64], const double *rhs, double x[8]) {
    HostEigen e;
    std::memcpy(e.A, S, 64 * sizeof(double));
    e.run(8);
    double threshold = 0;
    for (int i = 0; i < 8; i++) threshold += e.W[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < 8; j++) x[j] = 0;
    for (int i = 0; i < 8; i++) {
        double wi = e.W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        const double *u = e.V + 8 * i;
        if (rhs) {
            double s = 0;
            for (int j = 0; j < 8; j++) s += u[j] * rhs[j];
            s *= wi;
            for (int j = 0; j < 8; j++) x[j] = x[j] + s * u[j];
        } else {
            for (int j = 0; j < 8; j++) x[j] = x[j] + u[j] * (u[j] * wi);
        }
    }
}

// HomographyRefineCallback + LMSolverImpl::run (fundam.cpp, levmarq.cpp): residuals of the
// projective mapping with h[8] = 1, their Jacobian in the 8 free entries, and the
// damped Gauss-Newton iteration with OpenCV's gain-ratio rules.  Returns the iteration count.
class Refiner {
   public:
    Refiner(const float2 *src, const float2 *dst, int count) : M_(src), m_(dst), count_(count) {}

    int run(double h[8], int max_iters) {
        double x[8], xd[8], d[8], A[64], Ap[64], v[8], D[8];
        std::memcpy(x, h, sizeof x);
        double S = residuals(x, true, A, v);
        for (int i = 0; i < 8; i++) D[i] = A[9 * i];
        const double Rlo = 0.25, Rhi = 0.75;
        double lambda = 1, lc = 0.75;
        int iter = 0;
        for (;;) {
            std::memcpy(Ap, A, sizeof A);
            for (int i = 0; i < 8; i++) Ap[9 * i] += lambda * D[i];
            eig_apply(Ap, v, d);
            for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
            const double Sd = residuals(xd, false, nullptr, nullptr);
            double dS = 0;
            for (int i = 0; i < 8; i++) {
                double s = 0;
                for (int k = 0; k < 8; k++) s += A[8 * i + k] * d[k];
                dS += d[i] * (s * -1 + v[i] * 2);
            }
            const double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
            if (R > Rhi) {
                lambda *= 0.5;
                if (lambda < lc) lambda = 0;
            } else if (R < Rlo) {
                double t = 0;
                for (int i = 0; i < 8; i++) t += d[i] * v[i];
                double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
                nu = std::min(std::max(nu, 2.), 10.);
                if (lambda == 0) {
                    double inv_diag[8];
                    eig_apply(A, nullptr, inv_diag);
                    double maxval = DBL_EPSILON;
                    for (int i = 0; i < 8; i++) maxval = std::max(maxval, std::fabs(inv_diag[i]));
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            if (Sd < S) {
                std::memcpy(x, xd, sizeof x);
                S = residuals(x, true, A, v);
            }
            iter++;
            double dmax = 0;
            for (int i = 0; i < 8; i++) dmax = std::max(dmax, std::fabs(d[i]));
            if (!(iter < max_iters && dmax >= FLT_EPSILON && rmax_ >= FLT_EPSILON)) break;
        }
        std::memcpy(h, x, sizeof x);
        return iter;
    }

   private:
    // Sum of squared residuals at h.  With `with_normal`, also the residual vector's
    // largest magnitude (kept in rmax_), A = J^T J and v = J^T r, each entry summed over the
    // points in order (two rows per point).
    double residuals(const double h[8], bool with_normal, double *A, double *v) {
        double S = 0, rmax = 0;
        double acc[36] = {0}, vacc[8] = {0};
        for (int i = 0; i < count_; i++) {
            const double Mx = M_[i].x, My = M_[i].y;
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
            const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            const double ex = xi - m_[i].x, ey = yi - m_[i].y;
            S += ex * ex;
            S += ey * ey;
            if (!with_normal) continue;
            rmax = std::max(rmax, std::max(std::fabs(ex), std::fabs(ey)));
            const double jx[8] = {Mx * ww, My * ww, ww, 0., 0., 0., -Mx * ww * xi, -My * ww * xi};
            const double jy[8] = {0., 0., 0., Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            int q = 0;
            for (int a = 0; a < 8; a++) {
                for (int b = a; b < 8; b++, q++) {
                    acc[q] += jx[a] * jx[b];
                    acc[q] += jy[a] * jy[b];
                }
                vacc[a] += jx[a] * ex;
                vacc[a] += jy[a] * ey;
            }
        }
        if (with_normal) {
            rmax_ = rmax;
            int q = 0;
            for (int a = 0; a < 8; a++)
                for (int b = a; b < 8; b++, q++) A[8 * a + b] = A[8 * b + a] = acc[q];
            std::memcpy(v, vacc, sizeof vacc);
        }
        return S;
    }
    const float2 *M_, *m_;
    int count_;
    double rmax_ = 0;
};

}  // namespace

size_t homography_workspace_bytes(int n, int chunk) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return 2 * up((size_t)n * sizeof(float2)) + up((size_t)chunk * 4 * sizeof(int)) + up((size_t)chunk * 9 * sizeof(double)) +
           up((size_t)chunk) + up((size_t)chunk * sizeof(int)) + up((size_t)n);
}

// Returns hipSuccess or the failing HIP status; the outcome is in `info`.
hipError_t find_homography_ransac(hipStream_t st, void *workspace, const float *src_xy, const float *dst_xy, int n,
                                  double threshold, int max_iters, double confidence, double H[9], uint8_t *mask,
                                  sift_homography_info *info) {
    std::memset(info, 0, sizeof *info);
    for (int i = 0; i < 9; i++) H[i] = 0;
    if (mask && n > 0) std::memset(mask, 0, (size_t)n);
    if (n < 4) return hipSuccess;
    const float2 *src = reinterpret_cast<const float2 *>(src_xy), *dst = reinterpret_cast<const float2 *>(dst_xy);
    std::vector<uint8_t> best_mask((size_t)n, 0);
    bool found = false;
    if (n == 4) {
        if (fit_dlt_host(src, dst, 4, H) > 0) {
            std::fill(best_mask.begin(), best_mask.end(), 1);
            info->inliers = 4;
            found = true;
        }
    } else {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const int chunk_cap = HOMOGRAPHY_CHUNK;
        char *base = static_cast<char *>(workspace);
        float2 *d_src = reinterpret_cast<float2 *>(base);
        base += up((size_t)n * sizeof(float2));
        float2 *d_dst = reinterpret_cast<float2 *>(base);
        base += up((size_t)n * sizeof(float2));
        int *d_subsets = reinterpret_cast<int *>(base);
        base += up((size_t)chunk_cap * 4 * sizeof(int));
        double *d_models = reinterpret_cast<double *>(base);
        base += up((size_t)chunk_cap * 9 * sizeof(double));
        uint8_t *d_valid = reinterpret_cast<uint8_t *>(base);
        base += up((size_t)chunk_cap);
        int *d_counts = reinterpret_cast<int *>(base);
        base += up((size_t)chunk_cap * sizeof(int));
        uint8_t *d_mask = reinterpret_cast<uint8_t *>(base);

        hipError_t e;
#define H_TRY(expr)                    \
    do {                               \
        e = (expr);                    \
        if (e != hipSuccess) return e; \
    } while (0)
        H_TRY(hipMemcpyAsync(d_src, src, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        H_TRY(hipMemcpyAsync(d_dst, dst, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));

        const float thresh2 = (float)(threshold * threshold);
        Mwc rng{0xffffffffffffffffull};  // RNG rng((uint64)-1), ptsetreg.cpp
        int niters = std::max(max_iters, 1), max_good = 0, iter = 0;
        double best[9] = {0};
        std::vector<int> subsets((size_t)chunk_cap * 4), counts((size_t)chunk_cap);
        std::vector<double> models((size_t)chunk_cap * 9);
        std::vector<uint8_t> valid((size_t)chunk_cap);
        bool exhausted = false;
        while (iter < niters && !exhausted) {
            // the draws of the next iterations (they do not depend on the models)
            const int want = std::min(niters - iter, iter == 0 ? std::min(chunk_cap, HOMOGRAPHY_FIRST_CHUNK) : chunk_cap);
            int have = 0;
            for (; have < want; have++) {
                if (!draw_subset(src, dst, n, rng, 10000, &subsets[4 * (size_t)have])) {
                    exhausted = true;
                    break;
                }
            }
            if (have == 0) break;
            H_TRY(hipMemcpyAsync(d_subsets, subsets.data(), (size_t)have * 4 * sizeof(int), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_h_models, dim3((have + TEAMS - 1) / TEAMS), dim3(TEAM * TEAMS), 0, st, d_src, d_dst, d_subsets, have,
                               d_models, d_valid);
            hipLaunchKernelGGL(k_h_score, dim3((have + SCORE_WAVES - 1) / SCORE_WAVES), dim3(64 * SCORE_WAVES), 0, st, d_src, d_dst,
                               n, d_models, d_valid, have, thresh2, d_counts);
            H_TRY(hipGetLastError());
            H_TRY(hipMemcpyAsync(counts.data(), d_counts, (size_t)have * sizeof(int), hipMemcpyDeviceToHost, st));
            H_TRY(hipMemcpyAsync(valid.data(), d_valid, (size_t)have, hipMemcpyDeviceToHost, st));
            H_TRY(hipMemcpyAsync(models.data(), d_models, (size_t)have * 9 * sizeof(double), hipMemcpyDeviceToHost, st));
            H_TRY(hipStreamSynchronize(st));
            info->hypotheses_scored += have;
            // the sequential loop's bookkeeping over this chunk
            for (int k = 0; k < have && iter < niters; k++, iter++) {
                if (!valid[k]) continue;
                const int good = counts[k];
                if (good > std::max(max_good, 3)) {
                    std::memcpy(best, &models[9 * (size_t)k], sizeof best);
                    max_good = good;
                    niters = updated_iteration_budget(confidence, (double)(n - good) / n, 4, niters);
                    info->best_iteration = iter;
                }
            }
        }
        info->iterations = iter;
        if (max_good > 0) {
            std::memcpy(H, best, sizeof best);
            info->inliers = max_good;
            found = true;
            hipLaunchKernelGGL(k_h_mask, dim3((n + 255) / 256), dim3(256), 0, st, d_src, d_dst, n, to_float(best), thresh2, d_mask);
            H_TRY(hipGetLastError());
            H_TRY(hipMemcpyAsync(best_mask.data(), d_mask, (size_t)n, hipMemcpyDeviceToHost, st));
            H_TRY(hipStreamSynchronize(st));
        }
#undef H_TRY
    }
    if (found && n > 4) {
        // least squares over the inliers, then the Levenberg-Marquardt polish (fundam.cpp, findHomography)
        std::vector<float2> s1, d1;
        for (int i = 0; i < n; i++)
            if (best_mask[i]) {
                s1.push_back(src[i]);
                d1.push_back(dst[i]);
            }
        if (!s1.empty()) {
            fit_dlt_host(s1.data(), d1.data(), (int)s1.size(), H);
            Refiner refiner(s1.data(), d1.data(), (int)s1.size());
            info->lm_iterations = refiner.run(H, 10);
        }
    }
    if (found && mask) std::memcpy(mask, best_mask.data(), (size_t)n);
    info->found = found ? 1 : 0;
    return hipSuccess;
}

}  // namespace sa
