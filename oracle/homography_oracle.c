/*
 * CPU restatement of cv::findHomography(obj, scene, RANSAC, 3, mask) as the
 * reference calls it (src/main.cpp:105).  TEST INFRASTRUCTURE ONLY: nothing in
 * the product links or imports this file.
 *
 * PARITY UNPINNED.  The algorithm lives in OpenCV (calib3d), which the
 * reference does not vendor or pin (CMakeLists.txt:9) and which is not
 * installed here, and the reference holds no homography, inlier mask or
 * stitched output to compare with.  What follows restates the published
 * OpenCV 4.5 sources from their structure:
 *   modules/core/src/rand.cpp, core/include/opencv2/core/operations.hpp  RNG (multiply-with-carry)
 *   modules/calib3d/src/ptsetreg.cpp   RANSACPointSetRegistrator::run, getSubset, findInliers,
 *                                      RANSACUpdateNumIters
 *   modules/calib3d/src/fundam.cpp     HomographyEstimatorCallback (checkSubset, runKernel,
 *                                      computeError), HomographyRefineCallback, findHomography
 *   modules/core/src/lapack.cpp        JacobiImpl_ (cv::eigen), SVBkSb (solve/invert, DECOMP_EIG)
 *   modules/calib3d/src/levmarq.cpp    LMSolverImpl::run
 * Sums run in index order, one IEEE operation per written operation (the
 * file is compiled with -ffp-contract=off).
 */
#include "homography_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- cv::RNG ---------------------------------------------------------------- */
typedef struct {
    uint64_t state;
} orc_rng;

static unsigned rng_next(orc_rng *r) {
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
static int rng_uniform(orc_rng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

void orc_h_rng_sequence(uint64_t state, int n, uint32_t *out) {
    orc_rng r = {state ? state : 0xffffffffu};
    for (int i = 0; i < n; i++) out[i] = rng_next(&r);
}

/* ---- subset selection --------------------------------------------------------- */
static int collinear_last(const float *p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}

static double det3(const double a[9]) {
    return a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
}

static int check_subset(const float *s, const float *d, int count) {
    if (collinear_last(s, count) || collinear_last(d, count)) return 0;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int *t = tt[i];
            const double A[9] = {s[2 * t[0]], s[2 * t[0] + 1], 1., s[2 * t[1]], s[2 * t[1] + 1], 1., s[2 * t[2]], s[2 * t[2] + 1], 1.};
            const double B[9] = {d[2 * t[0]], d[2 * t[0] + 1], 1., d[2 * t[1]], d[2 * t[1] + 1], 1., d[2 * t[2]], d[2 * t[2] + 1], 1.};
            negative += det3(A) * det3(B) < 0;
        }
        if (negative != 0 && negative != 4) return 0;
    }
    return 1;
}

static int get_subset(const float *src, const float *dst, int count, orc_rng *rng, int max_attempts, int idx[4],
                      float ms1[8], float ms2[8]) {
    for (int iters = 0; iters < max_attempts; ++iters) {
        int i;
        for (i = 0; i < 4; ++i) {
            int idx_i;
            for (;;) {
                idx_i = rng_uniform(rng, 0, count);
                int seen = 0;
                for (int j = 0; j < i; j++) seen |= idx[j] == idx_i;
                if (!seen) break;
            }
            idx[i] = idx_i;
            ms1[2 * i] = src[2 * idx_i];
            ms1[2 * i + 1] = src[2 * idx_i + 1];
            ms2[2 * i] = dst[2 * idx_i];
            ms2[2 * i + 1] = dst[2 * idx_i + 1];
        }
        if (check_subset(ms1, ms2, i)) return 1;
    }
    return 0;
}

int orc_h_subsets(const float *src, const float *dst, int count, int n_iters, int32_t *idx_out) {
    orc_rng rng = {0xffffffffffffffffull};
    int k = 0;
    for (; k < n_iters; k++) {
        int idx[4];
        float a[8], b[8];
        if (!get_subset(src, dst, count, &rng, 10000, idx, a, b)) break;
        for (int j = 0; j < 4; j++) idx_out[4 * k + j] = idx[j];
    }
    return k;
}

/* ---- symmetric eigen-decomposition (Jacobi), eigenvectors in the rows of V ---- */
static double cv_hypot(double a, double b) {
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

#define ORC_MAXN 9
static void jacobi(double *A, double *W, double *V, int n) {
    const double eps = DBL_EPSILON;
    int i, j, k, m;
    int indR[ORC_MAXN], indC[ORC_MAXN];
    double mv = 0;
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) V[i * n + j] = 0;
        V[i * n + i] = 1;
    }
    const int max_iters = n * n * 30;
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) {
                const double val = fabs(A[n * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                const double val = fabs(A[n * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1)
        for (int iters = 0; iters < max_iters; iters++) {
            for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
                const double val = fabs(A[n * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (i = 1; i < n; i++) {
                const double val = fabs(A[n * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            const double p = A[n * k + l];
            if (fabs(p) <= eps) break;
            const double y = (W[l] - W[k]) * 0.5;
            double t = fabs(y) + cv_hypot(p, y);
            double s = cv_hypot(p, t);
            const double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[n * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define ROT(v0, v1) (a0 = (v0), b0 = (v1), (v0) = a0 * c - b0 * s, (v1) = a0 * s + b0 * c)
            for (i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
            for (i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
            for (i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
            for (i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
            for (j = 0; j < 2; j++) {
                const int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
                        const double val = fabs(A[n * idx + i]);
                        if (mv < val) mv = val, m = i;
                    }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                        const double val = fabs(A[n * i + idx]);
                        if (mv < val) mv = val, m = i;
                    }
                    indC[idx] = m;
                }
            }
        }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            double tmp = W[m];
            W[m] = W[k];
            W[k] = tmp;
            for (i = 0; i < n; i++) {
                tmp = V[n * m + i];
                V[n * m + i] = V[n * k + i];
                V[n * k + i] = tmp;
            }
        }
    }
}

/* ---- normalised DLT (HomographyEstimatorCallback::runKernel) ------------------- */
static void mat3_mul(const double a[9], const double b[9], double d[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

int orc_h_kernel(const float *M, const float *m, int count, double H[9]) {
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    int i;
    for (i = 0; i < count; i++) {
        cmx += m[2 * i];
        cmy += m[2 * i + 1];
        cMx += M[2 * i];
        cMy += M[2 * i + 1];
    }
    cmx /= count;
    cmy /= count;
    cMx /= count;
    cMy /= count;
    for (i = 0; i < count; i++) {
        smx += fabs(m[2 * i] - cmx);
        smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx);
        sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx;
    smy = count / smy;
    sMx = count / sMx;
    sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    memset(LtL, 0, sizeof LtL);
    for (i = 0; i < count; i++) {
        const double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        const double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[9 * j + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[9 * j + k] = LtL[9 * k + j];
    jacobi(LtL, W, V, 9);
    double Htemp[9], H0[9];
    mat3_mul(invHnorm, V + 72, Htemp);
    mat3_mul(Htemp, Hnorm2, H0);
    const double scale = 1. / H0[8];
    for (i = 0; i < 9; i++) H[i] = H0[i] * scale;
    return 1;
}

/* ---- reprojection error and inliers -------------------------------------------- */
int orc_h_inliers(const float *M, const float *m, int count, const double H[9], double thresh, uint8_t *mask) {
    const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    const float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < count; i++) {
        const float Mx = M[2 * i], My = M[2 * i + 1];
        const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
        const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - m[2 * i];
        const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - m[2 * i + 1];
        const float err = dx * dx + dy * dy;
        const int f = err <= t;
        if (mask) mask[i] = (uint8_t)f;
        nz += f;
    }
    return nz;
}

static int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = fmax(p, 0.);
    p = fmin(p, 1.);
    ep = fmax(ep, 0.);
    ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* ---- Levenberg-Marquardt refinement of the 8 free entries ----------------------- */
static void refine_compute(const float *M, const float *m, int count, const double h[8], double *err, double *J) {
    for (int i = 0; i < count; i++) {
        const double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[2 * i] = xi - m[2 * i];
        err[2 * i + 1] = yi - m[2 * i + 1];
        if (J) {
            double *p = J + 16 * (size_t)i;
            p[0] = Mx * ww;
            p[1] = My * ww;
            p[2] = ww;
            p[3] = p[4] = p[5] = 0.;
            p[6] = -Mx * ww * xi;
            p[7] = -My * ww * xi;
            p[8] = p[9] = p[10] = 0.;
            p[11] = Mx * ww;
            p[12] = My * ww;
            p[13] = ww;
            p[14] = -Mx * ww * yi;
            p[15] = -My * ww * yi;
        }
    }
}

static double norm_l2sqr(const double *v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return s;
}
static double norm_inf(const double *v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s = fmax(s, fabs(v[i]));
    return s;
}

/* A = J^T J (8x8), v = J^T r; rows of J in order */
static void normal_equations(const double *J, const double *r, int rows, double A[64], double v[8]) {
    for (int i = 0; i < 8; i++) {
        for (int j = i; j < 8; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += J[8 * (size_t)k + i] * J[8 * (size_t)k + j];
            A[8 * i + j] = s;
        }
        double s = 0;
        for (int k = 0; k < rows; k++) s += J[8 * (size_t)k + i] * r[k];
        v[i] = s;
    }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < i; j++) A[8 * i + j] = A[8 * j + i];
}

/* x = V^T diag(1/w) V b over the eigen-decomposition of the symmetric matrix S
 * (cv::solve / cv::invert with DECOMP_EIG: Jacobi then SVBkSb).  With b == NULL
 * the inverse (n x n) is produced instead. */
static void eig_solve(const double *S, const double *b, int n, double *x) {
    double A[64], W[8], V[64];
    memcpy(A, S, sizeof(double) * n * n);
    jacobi(A, W, V, n);
    const int nb = b ? 1 : n;
    for (int i = 0; i < n * nb; i++) x[i] = 0;
    double threshold = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        const double *u = V + n * i;
        if (b) {
            double s = 0;
            for (int j = 0; j < n; j++) s += u[j] * b[j];
            s *= wi;
            for (int j = 0; j < n; j++) x[j] = x[j] + s * u[j];
        } else {
            for (int j = 0; j < n; j++) {
                const double s = u[j] * wi;
                for (int k = 0; k < n; k++) x[n * k + j] += s * u[k];
            }
        }
    }
}

int orc_h_refine(const float *M, const float *m, int count, double H[9], int max_iters) {
    const int rows = 2 * count, lx = 8;
    double *r = malloc(sizeof(double) * rows), *rd = malloc(sizeof(double) * rows);
    double *J = malloc(sizeof(double) * 8 * (size_t)rows);
    double x[8], xd[8], A[64], Ap[64], v[8], d[8], temp_d[8], D[8];
    memcpy(x, H, sizeof x);
    refine_compute(M, m, count, x, r, J);
    double S = norm_l2sqr(r, rows);
    normal_equations(J, r, rows, A, v);
    for (int i = 0; i < lx; i++) D[i] = A[9 * i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof A);
        for (int i = 0; i < lx; i++) Ap[9 * i] += lambda * D[i];
        eig_solve(Ap, v, lx, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        refine_compute(M, m, count, xd, rd, NULL);
        const double Sd = norm_l2sqr(rd, rows);
        for (int i = 0; i < lx; i++) { /* temp_d = -A d + 2 v */
            double s = 0;
            for (int k = 0; k < lx; k++) s += A[8 * i + k] * d[k];
            temp_d[i] = s * -1 + v[i] * 2;
        }
        double dS = 0;
        for (int i = 0; i < lx; i++) dS += d[i] * temp_d[i];
        const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < lx; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = fmin(fmax(nu, 2.), 10.);
            if (lambda == 0) {
                eig_solve(A, NULL, lx, Ap);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) maxval = fmax(maxval, fabs(Ap[9 * i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof x);
            refine_compute(M, m, count, x, r, J);
            normal_equations(J, r, rows, A, v);
        }
        iter++;
        const int proceed = iter < max_iters && norm_inf(d, lx) >= FLT_EPSILON && norm_inf(r, rows) >= FLT_EPSILON;
        if (!proceed) break;
    }
    memcpy(H, x, sizeof x);
    free(r);
    free(rd);
    free(J);
    return iter;
}

/* ---- findHomography(src, dst, RANSAC, thresh, mask, maxIters, confidence) -------- */
int orc_find_homography(const float *src, const float *dst, int n, double thresh, int max_iters, double confidence,
                        double H[9], uint8_t *mask, orc_h_info *info) {
    orc_h_info local;
    if (!info) info = &local;
    memset(info, 0, sizeof *info);
    memset(H, 0, 9 * sizeof(double));
    if (mask) memset(mask, 0, n > 0 ? (size_t)n : 0);
    if (n < 4) return 0;
    uint8_t *best_mask = malloc((size_t)n), *cur_mask = malloc((size_t)n);
    int found = 0;
    if (n == 4) {
        if (orc_h_kernel(src, dst, 4, H) > 0) {
            memset(best_mask, 1, 4);
            found = 1;
            info->inliers = 4;
        }
    } else {
        orc_rng rng = {0xffffffffffffffffull};
        int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
        double best[9] = {0};
        for (iter = 0; iter < niters; iter++) {
            int idx[4];
            float ms1[8], ms2[8];
            double model[9];
            if (!get_subset(src, dst, n, &rng, 10000, idx, ms1, ms2)) {
                if (iter == 0) {
                    free(best_mask);
                    free(cur_mask);
                    return 0;
                }
                break;
            }
            if (orc_h_kernel(ms1, ms2, 4, model) <= 0) continue;
            const int good = orc_h_inliers(src, dst, n, model, thresh, cur_mask);
            if (good > (max_good > 3 ? max_good : 3)) {
                uint8_t *t = cur_mask;
                cur_mask = best_mask;
                best_mask = t;
                memcpy(best, model, sizeof best);
                max_good = good;
                niters = update_num_iters(confidence, (double)(n - good) / n, 4, niters);
                info->best_iteration = iter;
            }
        }
        info->iterations = iter;
        if (max_good > 0) {
            memcpy(H, best, sizeof best);
            found = 1;
            info->inliers = max_good;
        }
    }
    if (found && n > 4) {
        float *s1 = malloc(sizeof(float) * 2 * (size_t)n), *d1 = malloc(sizeof(float) * 2 * (size_t)n);
        int k = 0;
        for (int i = 0; i < n; i++)
            if (best_mask[i]) {
                s1[2 * k] = src[2 * i];
                s1[2 * k + 1] = src[2 * i + 1];
                d1[2 * k] = dst[2 * i];
                d1[2 * k + 1] = dst[2 * i + 1];
                k++;
            }
        if (k > 0) {
            orc_h_kernel(s1, d1, k, H); /* on failure H keeps the RANSAC model, as the caller's matrix does */
            info->lm_iterations = orc_h_refine(s1, d1, k, H, 10);
        }
        free(s1);
        free(d1);
    }
    if (found && mask) memcpy(mask, best_mask, (size_t)n);
    if (!found) memset(H, 0, 9 * sizeof(double));
    info->found = found;
    free(best_mask);
    free(cur_mask);
    return found;
}
