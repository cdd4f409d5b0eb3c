/*
 * sift_oracle.c -- CPU restatement of kannavkm/SIFT (see sift_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP engine, never the product.
 *
 * Citations "sift.cpp:N" are into /root/reference/src/sift.cpp.  "[OCV]" marks
 * behaviour of OpenCV 4.x primitives that the reference calls; OpenCV is not
 * vendored by the reference, so those parts restate OpenCV's published scalar
 * algorithms (parity unpinned beyond the 1112-keypoint known answer).
 *
 * Build with -ffp-contract=off: every floating-point operation below is one
 * IEEE-754 double operation in the written order, which is what the HIP
 * kernels reproduce.
 */
#define _GNU_SOURCE
#include "sift_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* sift.hpp:97-112 */
#define SCALES 3
#define BORDER 5
#define CONTRAST_THRESHOLD 0.04
#define SIGMA 1.6
#define ASSUMED_BLUR 0.5
#define IMAGES (SCALES + 3)
#define EIGEN_VALUE_RATIO 10.0
#define BINS 36
#define PEAK_RATIO 0.8
#define SCALE_FACTOR 1.5
#define RADIUS_FACTOR 3.0
#define PI 3.14159265358979323846
#define SCALE_MULTIPLIER 3.0
#define WINDOW_WIDTH 4
#define DESCRIPTOR_MAX 0.2
/* sift.cpp:17-18 */
#define EPS ((double)1e-7)
#define EPS2 ((double).0001)

struct orc_sift {
    int w, h;
    int octaves;
    double *onex;
    double *base;
    int *ow, *oh;
    double **gauss; /* [oct*IMAGES + lvl]  (sift.hpp:115 gauss_images) */
    double **dog;   /* [oct*(IMAGES-1) + lvl] (sift.hpp:114 images) */
    orc_keypoint *kp;
    int nkp, capkp;
    double *desc;
    orc_candidate *cand;
    int ncand, capcand;
    orc_stats *stats;
};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- tiny parallel-for over [begin,end), contiguous ranges per thread, the
 * way cv::parallel_for_ hands cv::Range stripes to the loop body ---------- */
typedef void (*range_fn)(void *ctx, int tid, int begin, int end);
typedef struct {
    range_fn fn;
    void *ctx;
    int tid, begin, end;
} pf_task;
static void *pf_thunk(void *p) {
    pf_task *t = (pf_task *)p;
    t->fn(t->ctx, t->tid, t->begin, t->end);
    return NULL;
}
static void parallel_for(int begin, int end, int threads, range_fn fn, void *ctx) {
    int n = end - begin;
    if (n <= 0) return;
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads == 1) {
        fn(ctx, 0, begin, end);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    pf_task *tk = (pf_task *)malloc(sizeof(pf_task) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        tk[t].fn = fn;
        tk[t].ctx = ctx;
        tk[t].tid = t;
        tk[t].begin = begin + (int)((int64_t)n * t / threads);
        tk[t].end = begin + (int)((int64_t)n * (t + 1) / threads);
        pthread_create(&th[t], NULL, pf_thunk, &tk[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(tk);
}

/* [OCV] cvRound: round half to even (lrint in the default rounding mode). */
static int cv_round(double v) { return (int)lrint(v); }

/* [OCV] borderInterpolate(p, len, BORDER_REFLECT_101): gfedcb|abcdefgh|gfedcba,
 * repeated until the index lands inside; the default border of GaussianBlur. */
static int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* sift.cpp:29 -- [OCV 4.x] cvtColor(BGR2GRAY) on CV_8U is integer arithmetic
 * with 15-bit coefficients B 3735, G 19235, R 9798 and a rounding shift. */
void orc_bgr_to_gray(const uint8_t *bgr, int w, int h, uint8_t *gray) {
    for (int64_t p = 0; p < (int64_t)w * h; p++) {
        int b = bgr[3 * p], g = bgr[3 * p + 1], r = bgr[3 * p + 2];
        gray[p] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    }
}

/* [OCV 4.x] getGaussianKernel(n, sigma, CV_64F) via getGaussianKernelBitExact:
 * t_i = exp(x_i^2 * (-0.125/sigma^2)) on doubled coordinates x_i = 1-n+2i for
 * the left half, sum = 2*sum(t_i) + 1, every tap multiplied by 1/sum, the
 * centre tap is 1/sum itself.  OpenCV evaluates exp in softfloat; libm's exp
 * is used here (difference at most an ulp of a tap: unpinned). */
static int gaussian_kernel(double sigma, double *k) {
    /* [OCV] GaussianBlur, non-8U depth: ksize = cvRound(sigma*4*2 + 1) | 1 */
    int n = cv_round(sigma * 4 * 2 + 1) | 1;
    int n2 = (n - 1) / 2;
    double scale2x = -0.125 / (sigma * sigma);
    double sum = 0.0;
    double t[64];
    int x = 1 - n;
    for (int i = 0; i < n2; i++, x += 2) {
        t[i] = exp((double)(x * x) * scale2x);
        sum += t[i];
    }
    sum *= 2.0;
    sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) {
        double v = t[i] * mul1;
        k[i] = v;
        k[n - 1 - i] = v;
    }
    k[n2] = 1.0 * mul1;
    return n;
}

int orc_gaussian_kernel(double sigma, double *coeffs) { return gaussian_kernel(sigma, coeffs); }

typedef struct {
    const double *src;
    double *dst;
    int w, h, n;
    const double *k;
    const int *xmap; /* w + n - 1 entries: reflect101(x - r) */
} blur_ctx;

/* [OCV] sepFilter2D row stage on CV_64F: RowFilter<double,double>: taps are
 * accumulated left to right, s = k[0]*S[0]; s += k[t]*S[t]. */
static void blur_rows(void *vc, int tid, int y0, int y1) {
    (void)tid;
    blur_ctx *c = (blur_ctx *)vc;
    for (int y = y0; y < y1; y++) {
        const double *S = c->src + (size_t)y * c->w;
        double *D = c->dst + (size_t)y * c->w;
        for (int x = 0; x < c->w; x++) {
            const int *m = c->xmap + x;
            double s = c->k[0] * S[m[0]];
            for (int t = 1; t < c->n; t++) s += c->k[t] * S[m[t]];
            D[x] = s;
        }
    }
}

/* [OCV] sepFilter2D column stage: SymmColumnFilter for a symmetric kernel:
 * s = k[c]*S[c] (+ delta 0); s += k[c+t]*(S[c+t] + S[c-t]) for t = 1..r. */
static void blur_cols(void *vc, int tid, int y0, int y1) {
    (void)tid;
    blur_ctx *c = (blur_ctx *)vc;
    int r = c->n / 2;
    const double *ky = c->k + r;
    for (int y = y0; y < y1; y++) {
        double *D = c->dst + (size_t)y * c->w;
        const double *Sc = c->src + (size_t)y * c->w;
        for (int x = 0; x < c->w; x++) D[x] = ky[0] * Sc[x] + 0.0;
        for (int t = 1; t <= r; t++) {
            const double *Sp = c->src + (size_t)reflect101(y + t, c->h) * c->w;
            const double *Sm = c->src + (size_t)reflect101(y - t, c->h) * c->w;
            double f = ky[t];
            for (int x = 0; x < c->w; x++) D[x] += f * (Sp[x] + Sm[x]);
        }
    }
}

/* cv::GaussianBlur(src, dst, Size(0,0), sigma, sigma) on CV_64F
 * (sift.cpp:42, sift.cpp:150). */
static void gaussian_blur(const double *src, double *dst, int w, int h, double sigma, int threads) {
    double k[64];
    int n = gaussian_kernel(sigma, k);
    int r = n / 2;
    int *xmap = (int *)malloc(sizeof(int) * (size_t)(w + n - 1));
    for (int i = 0; i < w + n - 1; i++) xmap[i] = reflect101(i - r, w);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)w * h);
    blur_ctx c1 = {src, tmp, w, h, n, k, xmap};
    parallel_for(0, h, threads, blur_rows, &c1);
    blur_ctx c2 = {tmp, dst, w, h, n, k, xmap};
    parallel_for(0, h, threads, blur_cols, &c2);
    free(tmp);
    free(xmap);
}

/* sift.cpp:40 -- [OCV] resize(INTER_LINEAR) to exactly twice the size:
 * source coordinate (d+0.5)/2-0.5, clamped at both ends.  Inputs are 8-bit
 * integers so every product and sum is exact in double. */
static void upsample2x(const double *src, int w, int h, double *dst) {
    int W = 2 * w, H = 2 * h;
    int *sx0 = (int *)malloc(sizeof(int) * (size_t)W);
    double *fx = (double *)malloc(sizeof(double) * (size_t)W);
    for (int dx = 0; dx < W; dx++) {
        double f = (dx + 0.5) * 0.5 - 0.5;
        int sx = (int)floor(f);
        f -= sx;
        if (sx < 0) { sx = 0; f = 0; }
        if (sx >= w - 1) { sx = w - 1; f = 0; }
        sx0[dx] = sx;
        fx[dx] = f;
    }
    double *row0 = (double *)malloc(sizeof(double) * (size_t)W);
    double *row1 = (double *)malloc(sizeof(double) * (size_t)W);
    for (int dy = 0; dy < H; dy++) {
        double f = (dy + 0.5) * 0.5 - 0.5;
        int sy = (int)floor(f);
        f -= sy;
        if (sy < 0) { sy = 0; f = 0; }
        if (sy >= h - 1) { sy = h - 1; f = 0; }
        int sy1 = sy + 1 < h ? sy + 1 : h - 1;
        const double *S0 = src + (size_t)sy * w, *S1 = src + (size_t)sy1 * w;
        for (int dx = 0; dx < W; dx++) {
            int a = sx0[dx], b = a + 1 < w ? a + 1 : w - 1;
            row0[dx] = S0[a] * (1.0 - fx[dx]) + S0[b] * fx[dx];
            row1[dx] = S1[a] * (1.0 - fx[dx]) + S1[b] * fx[dx];
        }
        double *D = dst + (size_t)dy * W;
        for (int dx = 0; dx < W; dx++) D[dx] = row0[dx] * (1.0 - f) + row1[dx] * f;
    }
    free(sx0); free(fx); free(row0); free(row1);
}

/* sift_handler::sift_handler, sift.cpp:26-46 (after cvtColor). */
orc_sift *orc_create_gray(const uint8_t *gray, int w, int h, int stride) {
    orc_sift *s = (orc_sift *)calloc(1, sizeof(orc_sift));
    s->w = w;
    s->h = h;
    s->onex = (double *)malloc(sizeof(double) * (size_t)w * h); /* sift.cpp:30,33 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) s->onex[(size_t)y * w + x] = (double)gray[(size_t)y * stride + x];
    /* sift.cpp:36 */
    s->octaves = (int)round(log2((double)(w < h ? w : h)));
    /* sift.cpp:40 */
    double *interp = (double *)malloc(sizeof(double) * (size_t)4 * w * h);
    upsample2x(s->onex, w, h, interp);
    /* sift.cpp:41-42 */
    double diff = fmax(sqrt(pow(SIGMA, 2) - 4 * pow(ASSUMED_BLUR, 2)), 0.1);
    s->base = (double *)malloc(sizeof(double) * (size_t)4 * w * h);
    gaussian_blur(interp, s->base, 2 * w, 2 * h, diff, 1);
    free(interp);
    s->ow = (int *)calloc((size_t)s->octaves + 1, sizeof(int));
    s->oh = (int *)calloc((size_t)s->octaves + 1, sizeof(int));
    s->gauss = (double **)calloc((size_t)s->octaves * IMAGES + 1, sizeof(double *));
    s->dog = (double **)calloc((size_t)s->octaves * (IMAGES - 1) + 1, sizeof(double *));
    s->stats = (orc_stats *)calloc((size_t)s->octaves * 3 + 1, sizeof(orc_stats));
    return s;
}

void orc_destroy(orc_sift *s) {
    if (!s) return;
    for (int i = 0; i < s->octaves * IMAGES; i++) free(s->gauss[i]);
    for (int i = 0; i < s->octaves * (IMAGES - 1); i++) free(s->dog[i]);
    free(s->gauss); free(s->dog); free(s->ow); free(s->oh);
    free(s->onex); free(s->base); free(s->kp); free(s->desc); free(s->cand); free(s->stats);
    free(s);
}

/* sift_handler::gen_gaussian_images, sift.cpp:132-156. */
void orc_gen_gaussian_images(orc_sift *s, int blur_threads) {
    double k = pow(2, 1.0 / SCALES); /* :134 */
    double kernel[IMAGES];
    double prev = kernel[0] = SIGMA;
    for (int i = 1; i < IMAGES; i++) { /* :137-141 */
        double now = prev * k;
        kernel[i] = sqrt(pow(now, 2) - pow(prev, 2));
        prev = now;
    }
    int w = 2 * s->w, h = 2 * s->h;
    double *temp = (double *)malloc(sizeof(double) * (size_t)w * h); /* :143 base.clone() */
    memcpy(temp, s->base, sizeof(double) * (size_t)w * h);
    for (int o = 0; o < s->octaves; o++) {
        s->ow[o] = w;
        s->oh[o] = h;
        double **G = s->gauss + (size_t)o * IMAGES;
        G[0] = temp; /* :148 */
        for (int j = 1; j < IMAGES; j++) { /* :149-151 */
            G[j] = (double *)malloc(sizeof(double) * (size_t)w * h);
            gaussian_blur(G[j - 1], G[j], w, h, kernel[j], blur_threads);
        }
        /* :152-153 -- [OCV] resize(fx=fy=0.5, INTER_NEAREST): dsize =
         * cvRound(dim*0.5) (half to even), source index min(floor(d*2), dim-1). */
        int nw = cv_round(w * 0.5), nh = cv_round(h * 0.5);
        const double *src = G[IMAGES - 3];
        temp = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1) * (nh > 0 ? nh : 1));
        for (int y = 0; y < nh; y++) {
            int sy = 2 * y < h - 1 ? 2 * y : h - 1;
            for (int x = 0; x < nw; x++) {
                int sx = 2 * x < w - 1 ? 2 * x : w - 1;
                temp[(size_t)y * nw + x] = src[(size_t)sy * w + sx];
            }
        }
        w = nw;
        h = nh;
    }
    free(temp);
}

/* sift_handler::gen_dog_images, sift.cpp:161-177. */
void orc_gen_dog_images(orc_sift *s) {
    for (int o = 0; o < s->octaves; o++) {
        size_t n = (size_t)s->ow[o] * s->oh[o];
        for (int j = 1; j < IMAGES; j++) { /* :172-174 cv::subtract(octave[j], octave[j-1]) */
            const double *a = s->gauss[o * IMAGES + j], *b = s->gauss[o * IMAGES + j - 1];
            double *d = (double *)malloc(sizeof(double) * n);
            for (size_t p = 0; p < n; p++) d[p] = a[p] - b[p];
            s->dog[o * (IMAGES - 1) + j - 1] = d;
        }
    }
}

/* per-thread accumulation (cv::TLSDataAccumulator, sift.cpp:184) */
typedef struct {
    orc_keypoint *kp;
    int n, cap;
    orc_candidate *cand;
    int ncand, capcand;
    orc_stats *stats;
} tls_t;

static void tls_push_kp(tls_t *t, orc_keypoint k) {
    if (t->n == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 256;
        t->kp = (orc_keypoint *)realloc(t->kp, sizeof(orc_keypoint) * (size_t)t->cap);
    }
    t->kp[t->n++] = k;
}
static void tls_push_cand(tls_t *t, orc_candidate c) {
    if (t->ncand == t->capcand) {
        t->capcand = t->capcand ? 2 * t->capcand : 256;
        t->cand = (orc_candidate *)realloc(t->cand, sizeof(orc_candidate) * (size_t)t->capcand);
    }
    t->cand[t->ncand++] = c;
}

typedef struct {
    orc_sift *s;
    int oct, img;
    tls_t *tls;
} scan_ctx;

/* get_pixel_cube, sift.cpp:251-259: cube[k][a][b] = images[oct][img-1+k](i-1+a, j-1+b). */
static void get_pixel_cube(const orc_sift *s, int oct, int img, int i, int j, double cube[3][3][3]) {
    int w = s->ow[oct];
    for (int k = 0; k < 3; k++) {
        const double *D = s->dog[oct * (IMAGES - 1) + img - 1 + k];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) cube[k][a][b] = D[(size_t)(i - 1 + a) * w + (j - 1 + b)];
    }
}

/* is_pixel_extremum, sift.cpp:264-281. */
static int is_pixel_extremum(double cube[3][3][3]) {
    int is_maximum = 1, is_minimum = 1;
    double threshold = floor(0.5 * CONTRAST_THRESHOLD / SCALES * 255); /* :266 */
    double c = cube[1][1][1];
    if (fabs(c) <= threshold) return 0; /* :268 */
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                is_maximum &= c >= cube[k][i][j];
                is_minimum &= c <= cube[k][i][j];
            }
    return ((c < 0) & is_minimum) | ((c > 0) & is_maximum); /* :280 */
}

/* [OCV] hal::LU64f (LUImpl<double>, eps = DBL_EPSILON*100): in-place LU with
 * partial pivoting on the m x m matrix A and one right-hand side b. */
static int lu_solve3(double A[3][3], double b[3]) {
    const int m = 3;
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j][i]) > fabs(A[k][i])) k = j;
        if (fabs(A[k][i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) { double t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[i][i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j][i] * d;
            for (int kk = i + 1; kk < m; kk++) A[j][kk] += alpha * A[i][kk];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double sacc = b[i];
        for (int k = i + 1; k < m; k++) sacc -= A[i][k] * b[k];
        b[i] = sacc / A[i][i];
    }
    return 1;
}

/* [OCV] cv::solve(H, g, x, DECOMP_NORMAL) (sift.cpp:332): forms the normal
 * equations N = H^T H (mulTransposed: upper triangle as sequential sums over
 * k, mirrored) and r = H^T g (gemm with GEMM_1_T, sequential over k), then
 * solves by LU. */
static int solve_normal3(const double H[3][3], const double g[3], double x[3]) {
    double N[3][3], r[3];
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) {
            double sacc = 0;
            for (int k = 0; k < 3; k++) sacc += H[k][i] * H[k][j];
            N[i][j] = sacc;
            N[j][i] = sacc;
        }
    for (int i = 0; i < 3; i++) {
        double sacc = 0;
        for (int k = 0; k < 3; k++) sacc += H[k][i] * g[k];
        r[i] = sacc;
    }
    if (!lu_solve3(N, r)) return 0;
    x[0] = r[0]; x[1] = r[1]; x[2] = r[2];
    return 1;
}

/* localize_extrema, sift.cpp:312-387.  Returns the layer index, 0 when
 * cv::solve fails (:333-335) and -1 on rejection. */
static int localize_extrema(const orc_sift *s, int oct, int img, int i, int j, orc_keypoint *kpt, orc_stats *st) {
    const int attempts = 5;
    int w = s->ow[oct], h = s->oh[oct];
    int attempt;
    double cube[3][3][3], grad[3], hess[3][3], res[3];
    for (attempt = 0; attempt < attempts; attempt++) {
        get_pixel_cube(s, oct, img, i, j, cube);
        /* :324-326 -- [OCV] Mat /= 255.0 multiplies by the reciprocal */
        const double inv = 1. / 255.0;
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) cube[k][a][b] = cube[k][a][b] * inv;
        /* get_gradient, :286-292 */
        grad[0] = 0.5 * (cube[1][1][2] - cube[1][1][0]);
        grad[1] = 0.5 * (cube[1][2][1] - cube[1][0][1]);
        grad[2] = 0.5 * (cube[2][1][1] - cube[0][1][1]);
        /* get_hessian, :297-310 */
        hess[0][0] = cube[1][1][2] - 2 * cube[1][1][1] + cube[1][1][0];
        hess[1][1] = cube[1][2][1] - 2 * cube[1][1][1] + cube[1][0][1];
        hess[2][2] = cube[2][1][1] - 2 * cube[1][1][1] + cube[0][1][1];
        hess[0][1] = hess[1][0] = 0.25 * (cube[1][2][2] - cube[1][2][0] - cube[1][0][2] + cube[1][0][0]);
        hess[0][2] = hess[2][0] = 0.25 * (cube[2][1][2] - cube[2][1][0] - cube[0][1][2] + cube[0][1][0]);
        hess[1][2] = hess[2][1] = 0.25 * (cube[2][2][1] - cube[2][0][1] - cube[0][2][1] + cube[0][0][1]);
        if (!solve_normal3(hess, grad, res)) { /* :332-335 */
            st->singular++;
            return 0;
        }
        res[0] *= -1; res[1] *= -1; res[2] *= -1; /* :336 */
        if (fabs(res[0]) < 0.5 && fabs(res[1]) < 0.5 && fabs(res[2]) < 0.5) break; /* :339 */
        j += (int)round(res[0]); /* :342-344, std::round: half away from zero */
        i += (int)round(res[1]);
        img += (int)round(res[2]);
        if (i < BORDER || i >= h - BORDER || j < BORDER || j >= w - BORDER || img < 1 || img > SCALES) { /* :347 */
            st->out_of_bounds++;
            return -1;
        }
    }
    if (attempt == 5) { /* :356 */
        st->no_convergence++;
        return -1;
    }
    /* :360 -- Mat::dot: sequential sum of products */
    double dot = 0;
    dot += grad[0] * res[0];
    dot += grad[1] * res[1];
    dot += grad[2] * res[2];
    double value = cube[1][1][1] + 0.5 * dot;
    if (fabs(value) * SCALES >= CONTRAST_THRESHOLD) { /* :363 */
        double hess_trace = hess[0][0] + hess[1][1];                             /* :365 */
        double hess_det = hess[0][0] * hess[1][1] - hess[0][1] * hess[1][0];     /* :366 [OCV] det2 */
        if (hess_det <= 0) { /* :367 */
            st->det_nonpositive++;
            return -1;
        }
        if ((hess_trace * hess_trace) * EIGEN_VALUE_RATIO < pow(EIGEN_VALUE_RATIO + 1, 2) * hess_det) { /* :374 */
            int keypt_octave = oct + (1 << 8) * img + (1 << 16) * (int)round((res[2] + 0.5) * 255); /* :375 */
            double keypt_pt_x = (j + res[0]) * (1 << oct);                                            /* :376 */
            double keypt_pt_y = (i + res[1]) * (1 << oct);                                            /* :377 */
            double keypt_size = SIGMA * (pow(2, img + res[2] / SCALES)) * (1 << (oct + 1));           /* :378 */
            double keypt_response = fabs(value);                                                      /* :379 */
            kpt->x = (float)keypt_pt_x; /* :380 cv::KeyPoint(float...) narrows */
            kpt->y = (float)keypt_pt_y;
            kpt->size = (float)keypt_size;
            kpt->angle = -1.0F;
            kpt->response = (float)keypt_response;
            kpt->octave = keypt_octave;
            kpt->class_id = -1;
            st->localized++;
            return img;
        }
        st->edge_like++;
        return -1;
    }
    st->low_contrast++;
    return -1;
}

/* get_keypoint_orientations, sift.cpp:393-448. */
static void get_keypoint_orientations(const orc_sift *s, int oct, int img, const orc_keypoint *kpt, tls_t *out,
                                      orc_stats *st) {
    int w = s->ow[oct], h = s->oh[oct];
    const double *L = s->gauss[oct * IMAGES + img];
    double hist[BINS], smooth[BINS];
    for (int b = 0; b < BINS; b++) hist[b] = 0, smooth[b] = 0;
    int base_x = (int)round(kpt->x / (double)(1 << oct)); /* :398 */
    int base_y = (int)round(kpt->y / (double)(1 << oct)); /* :399 */
    double scale = SCALE_FACTOR * kpt->size / (double)(1 << (oct + 1)); /* :401 */
    int radius = (int)round(scale * RADIUS_FACTOR);                     /* :402 */
    double weight_factor = -0.5 / (scale * scale);                      /* :403 */
    for (int i = -radius; i <= radius; i++) {
        if (base_y + i > 0 && base_y + i < h - 1) {
            for (int j = -radius; j <= radius; j++) {
                if (base_x + j > 0 && base_x + j < w - 1) {
                    double dx = L[(size_t)(base_y + i) * w + base_x + j + 1] - L[(size_t)(base_y + i) * w + base_x + j - 1];
                    double dy = L[(size_t)(base_y + i - 1) * w + base_x + j] - L[(size_t)(base_y + i + 1) * w + base_x + j];
                    double mag = sqrt(dx * dx + dy * dy);
                    double orientation = (atan2(dy, dx) * 180) / PI; /* rad2deg, :20 */
                    int index = ((int)round((orientation * BINS) / 360) % BINS + BINS) % BINS; /* :416 */
                    hist[index] += exp(weight_factor * (i * i + j * j)) * mag;                /* :417 */
                }
            }
        }
    }
#define CIRC(i) (((i) + BINS) % BINS)
    for (int i = 0; i < BINS; i++) /* :426-429 */
        smooth[i] = (6 * hist[i] + 4 * (hist[CIRC(i - 1)] + hist[CIRC(i + 1)]) + hist[CIRC(i - 2)] + hist[CIRC(i + 2)]) / 16.;
    double max_orientation = smooth[0]; /* :432 */
    for (int i = 1; i < BINS; i++)
        if (smooth[i] > max_orientation) max_orientation = smooth[i];
    for (int i = 0; i < BINS; i++) {
        double l = smooth[CIRC(i - 1)], r = smooth[CIRC(i + 1)];
        if (smooth[i] > l && smooth[i] > r) { /* :435 */
            double peak = smooth[i];
            if (peak >= PEAK_RATIO * max_orientation) { /* :437 */
                double interpolated_index = fmod((i + 0.5 * (l - r) / (l + r - 2 * peak)), (double)BINS); /* :438 */
                double orientation = 360 - interpolated_index * 360 / BINS;                               /* :439 */
                if (fabs(360 - orientation) < EPS) orientation = 0;                                       /* :440 */
                orc_keypoint nk = *kpt; /* :443 */
                nk.angle = (float)orientation;
                tls_push_kp(out, nk);
                st->oriented++;
            }
        }
    }
#undef CIRC
}

/* scale_space_extrema_parallel::operator(), sift.cpp:227-246. */
static void scan_rows(void *vc, int tid, int begin, int end) {
    scan_ctx *c = (scan_ctx *)vc;
    orc_sift *s = c->s;
    tls_t *tls = &c->tls[tid];
    orc_stats *st = &tls->stats[c->oct * 3 + c->img - 1];
    int w = s->ow[c->oct];
    double cube[3][3][3];
    for (int i = begin; i < end; i++) {
        for (int j = BORDER; j < w - BORDER; j++) {
            get_pixel_cube(s, c->oct, c->img, i, j, cube);
            if (is_pixel_extremum(cube)) {
                orc_candidate cd = {c->oct, c->img, i, j};
                tls_push_cand(tls, cd);
                st->candidates++;
                orc_keypoint kpt;
                memset(&kpt, 0, sizeof kpt); /* default cv::KeyPoint: size 0 */
                kpt.angle = -1;
                kpt.class_id = -1;
                int image_index = localize_extrema(s, c->oct, c->img, i, j, &kpt, st);
                if (image_index < 0) continue; /* :239 */
                /* image_index 0 (solve failed) runs orientation on a default
                 * keypoint: radius 0 at (0,0) fails base_y + i > 0 and emits nothing */
                get_keypoint_orientations(s, c->oct, image_index, &kpt, tls, st);
            }
        }
    }
}

static int cmp_cand(const void *a, const void *b) {
    const orc_candidate *x = (const orc_candidate *)a, *y = (const orc_candidate *)b;
    if (x->oct != y->oct) return x->oct < y->oct ? -1 : 1;
    if (x->img != y->img) return x->img < y->img ? -1 : 1;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    if (x->j != y->j) return x->j < y->j ? -1 : 1;
    return 0;
}

/* sift_handler::gen_scale_space_extrema, sift.cpp:183-197. */
void orc_gen_scale_space_extrema(orc_sift *s, int scan_threads) {
    if (scan_threads < 1) scan_threads = 1;
    tls_t *tls = (tls_t *)calloc((size_t)scan_threads, sizeof(tls_t));
    for (int t = 0; t < scan_threads; t++) tls[t].stats = (orc_stats *)calloc((size_t)s->octaves * 3 + 1, sizeof(orc_stats));
    for (int oct = 0; oct < s->octaves; oct++) {
        for (int img = 1; img < IMAGES - 1 - 1; img++) { /* :186: 1 .. images[oct].size()-2 */
            scan_ctx c = {s, oct, img, tls};
            parallel_for(BORDER, s->oh[oct] - BORDER, scan_threads, scan_rows, &c); /* :188 */
        }
    }
    for (int t = 0; t < scan_threads; t++) { /* :192-196 gather */
        for (int k = 0; k < tls[t].n; k++) {
            if (s->nkp == s->capkp) {
                s->capkp = s->capkp ? 2 * s->capkp : 1024;
                s->kp = (orc_keypoint *)realloc(s->kp, sizeof(orc_keypoint) * (size_t)s->capkp);
            }
            s->kp[s->nkp++] = tls[t].kp[k];
        }
        for (int k = 0; k < tls[t].ncand; k++) {
            if (s->ncand == s->capcand) {
                s->capcand = s->capcand ? 2 * s->capcand : 1024;
                s->cand = (orc_candidate *)realloc(s->cand, sizeof(orc_candidate) * (size_t)s->capcand);
            }
            s->cand[s->ncand++] = tls[t].cand[k];
        }
        for (int q = 0; q < s->octaves * 3; q++) {
            orc_stats *a = &s->stats[q], *b = &tls[t].stats[q];
            a->candidates += b->candidates; a->out_of_bounds += b->out_of_bounds;
            a->no_convergence += b->no_convergence; a->low_contrast += b->low_contrast;
            a->det_nonpositive += b->det_nonpositive; a->edge_like += b->edge_like;
            a->singular += b->singular; a->localized += b->localized; a->oriented += b->oriented;
        }
        free(tls[t].kp); free(tls[t].cand); free(tls[t].stats);
    }
    free(tls);
    if (s->ncand) qsort(s->cand, (size_t)s->ncand, sizeof(orc_candidate), cmp_cand);
}

/* comparator of sift.cpp:204-213 as a three-way compare */
static int cmp_kp(const void *a, const void *b) {
    const orc_keypoint *k1 = (const orc_keypoint *)a, *k2 = (const orc_keypoint *)b;
    if (k1->x != k2->x) return k1->x < k2->x ? -1 : 1;
    if (k1->y != k2->y) return k1->y < k2->y ? -1 : 1;
    if (k1->size != k2->size) return k1->size > k2->size ? -1 : 1;
    if (k1->angle != k2->angle) return k1->angle < k2->angle ? -1 : 1;
    if (k1->response != k2->response) return k1->response > k2->response ? -1 : 1;
    if (k1->octave != k2->octave) return k1->octave > k2->octave ? -1 : 1;
    if (k1->class_id != k2->class_id) return k1->class_id > k2->class_id ? -1 : 1;
    return 0;
}

/* sift_handler::clean_keypoints, sift.cpp:202-225. */
void orc_clean_keypoints(orc_sift *s) {
    if (s->nkp == 0) return;
    /* std::sort; elements that compare equal are identical in all 7 fields */
    qsort(s->kp, (size_t)s->nkp, sizeof(orc_keypoint), cmp_kp);
    /* std::unique with the predicate of :214-217: compares against the last
     * KEPT element; float differences widened to double for the comparison;
     * pt.x is tested twice and pt.y never. */
    int last = 0;
    for (int i = 1; i < s->nkp; i++) {
        const orc_keypoint *I = &s->kp[last], *J = &s->kp[i];
        int differ = (double)fabsf(I->x - J->x) > EPS2 || (double)fabsf(I->x - J->x) > EPS2 ||
                     (double)fabsf(I->size - J->size) > EPS2 || (double)fabsf(I->angle - J->angle) > EPS2;
        if (differ) s->kp[++last] = s->kp[i];
    }
    s->nkp = last + 1;
    for (int i = 0; i < s->nkp; i++) { /* :219-223 */
        orc_keypoint *k = &s->kp[i];
        k->x = (float)(k->x * 0.5);
        k->y = (float)(k->y * 0.5);
        k->size = (float)(k->size * 0.5);
        k->octave = (k->octave & ~255) | ((k->octave - 1) & 255);
    }
}

typedef struct {
    orc_sift *s;
} desc_ctx;

/* get_descriptors_parallel::operator(), sift.cpp:464-584. */
static void desc_range(void *vc, int tid, int begin, int end) {
    (void)tid;
    orc_sift *s = ((desc_ctx *)vc)->s;
    size_t cap = 0;
    double *rows = NULL, *cols = NULL, *magnitudes = NULL, *orientations = NULL;
    for (int it = begin; it < end; it++) {
        orc_keypoint kpt = s->kp[it];
        int octave = kpt.octave & 255;        /* :469 */
        int layer = (kpt.octave >> 8) & 255;  /* :470 */
        if (octave >= 128) octave |= -128;    /* :471-473 */
        double scale = octave >= 0 ? (1.0 / (double)(1 << octave)) : ((double)(1 << -octave)); /* :474 */
        const double *image = s->gauss[(octave + 1) * IMAGES + layer];                           /* :476 */
        int width = s->ow[octave + 1], height = s->oh[octave + 1];
        int pt_x = (int)round(scale * kpt.x); /* :479 */
        int pt_y = (int)round(scale * kpt.y); /* :480 */
        double angle = 360 - kpt.angle;       /* :481 */
        double cosv = cos(((double)(angle)*PI) / 180); /* :482, deg2rad :19 */
        double sinv = sin(((double)(angle)*PI) / 180); /* :483 */
        double weight_multiplier = -0.5 / (pow(0.5 * WINDOW_WIDTH, 2)); /* :484 */
        const int bins = 8;
        const double bins_per_degree = 8 * 1.0 / 360.0; /* :487 */
        double hist_width = SCALE_MULTIPLIER * 0.5 * scale * kpt.size; /* :491 */
        int half_width = (int)fmin(round(hist_width * (WINDOW_WIDTH + 1) / sqrt(2.0)),
                                   sqrt(pow(height, 2) + pow(width, 2))); /* :492-493 */
        size_t need = (size_t)(2 * half_width + 1) * (size_t)(2 * half_width + 1);
        if (need > cap) {
            cap = need;
            rows = (double *)realloc(rows, sizeof(double) * cap);
            cols = (double *)realloc(cols, sizeof(double) * cap);
            magnitudes = (double *)realloc(magnitudes, sizeof(double) * cap);
            orientations = (double *)realloc(orientations, sizeof(double) * cap);
        }
        size_t n = 0;
        for (int i = -half_width; i <= half_width; i++) {
            for (int j = -half_width; j <= half_width; j++) {
                double row_rotation = sinv * j + cosv * i; /* :497 */
                double col_rotation = sinv * i - cosv * j; /* :498 */
                double bin_row = (row_rotation / hist_width) + 0.5 * (WINDOW_WIDTH - 1); /* :500 */
                double bin_col = (col_rotation / hist_width) + 0.5 * (WINDOW_WIDTH - 1); /* :501 */
                if (bin_row > -1 && bin_col > -1 && bin_row < WINDOW_WIDTH && bin_col < WINDOW_WIDTH) { /* :503 */
                    int win_row = pt_y + i, win_col = pt_x + j;
                    if (win_row > 0 && win_col > 0 && win_row < height - 1 && win_col < width - 1) { /* :506 */
                        double dx = image[(size_t)win_row * width + win_col + 1] - image[(size_t)win_row * width + win_col - 1];
                        double dy = image[(size_t)(win_row - 1) * width + win_col] - image[(size_t)(win_row + 1) * width + win_col];
                        double mag = sqrt(dx * dx + dy * dy);                       /* :509, pow(x,2) == x*x */
                        double orient = fmod((atan2(dy, dx) * 180) / PI, 360.0);   /* :510 */
                        double e1 = row_rotation / hist_width, e2 = col_rotation / hist_width;
                        double exponent = e1 * e1 + e2 * e2;                        /* :512-513 */
                        double weight = exp(weight_multiplier * exponent);          /* :514 */
                        rows[n] = bin_row;
                        cols[n] = bin_col;
                        magnitudes[n] = mag * weight;
                        orientations[n] = (orient - angle) * bins_per_degree; /* :519 */
                        n++;
                    }
                }
            }
        }
        double tensor[WINDOW_WIDTH + 2][WINDOW_WIDTH + 2][8]; /* :524-525 */
        memset(tensor, 0, sizeof tensor);
        for (size_t l = 0; l < n; l++) { /* :527-553 */
            int row_bin = (int)floor(rows[l]);
            int col_bin = (int)floor(cols[l]);
            int orient_bin = (int)floor(orientations[l]);
            double row_bin_pr = rows[l] - row_bin;
            double col_bin_pr = cols[l] - col_bin;
            double orient_bin_pr = orientations[l] - orient_bin;
            if (orient_bin < 0) orient_bin += bins;
            if (orient_bin >= bins) orient_bin -= bins;
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int k = 0; k < 2; k++) {
                        double c = magnitudes[l];
                        c *= (!i ? (1 - row_bin_pr) : (row_bin_pr));
                        c *= (!j ? (1 - col_bin_pr) : (col_bin_pr));
                        c *= (!k ? (1 - orient_bin_pr) : (orient_bin_pr));
                        tensor[row_bin + 1 + i][col_bin + 1 + j][((orient_bin + k) % bins + bins) % bins] += c;
                    }
        }
        double *d = s->desc + (size_t)it * 128;
        int q = 0;
        for (int i = 1; i <= WINDOW_WIDTH; i++) /* :557-563 */
            for (int j = 1; j <= WINDOW_WIDTH; j++)
                for (int k = 0; k < bins; k++) d[q++] = tensor[i][j][k];
        /* :564-565 std::inner_product(..., 0.0L): products in double, sum in long double */
        long double acc = 0.0L;
        for (int k = 0; k < 128; k++) acc = acc + (long double)(d[k] * d[k]);
        double norm = (double)sqrtl(acc);
        double thresh = norm * DESCRIPTOR_MAX; /* :566 */
        for (int k = 0; k < 128; k++) {        /* :567-573 */
            double I = d[k];
            I = I > thresh ? thresh : I;
            I /= fmax(norm, EPS);
            I = round(I * 512);
            I = I < 0 ? 0 : I;
            I = I > 255 ? 255 : I;
            d[k] = I;
        }
    }
    free(rows); free(cols); free(magnitudes); free(orientations);
}

/* sift_handler::get_descriptors, sift.cpp:450-462. */
void orc_get_descriptors(orc_sift *s, int scan_threads) {
    free(s->desc);
    s->desc = (double *)calloc((size_t)(s->nkp > 0 ? s->nkp : 1) * 128, sizeof(double));
    desc_ctx c = {s};
    parallel_for(0, s->nkp, scan_threads, desc_range, &c);
}

/* sift_handler::exec, sift.cpp:67-98 (drawing and file output omitted). */
void orc_exec(orc_sift *s, int scan_threads, int blur_threads, double *timings) {
    double t0 = now_s();
    orc_gen_gaussian_images(s, blur_threads);
    double t1 = now_s();
    orc_gen_dog_images(s);
    double t2 = now_s();
    orc_gen_scale_space_extrema(s, scan_threads);
    double t3 = now_s();
    orc_clean_keypoints(s);
    double t4 = now_s();
    orc_get_descriptors(s, scan_threads);
    double t5 = now_s();
    if (timings) {
        timings[0] = t1 - t0; timings[1] = t2 - t1; timings[2] = t3 - t2;
        timings[3] = t4 - t3; timings[4] = t5 - t4;
    }
}

int orc_octaves(const orc_sift *s) { return s->octaves; }
void orc_level_dims(const orc_sift *s, int oct, int *w, int *h) { *w = s->ow[oct]; *h = s->oh[oct]; }
const double *orc_onex(const orc_sift *s) { return s->onex; }
const double *orc_base(const orc_sift *s) { return s->base; }
const double *orc_gauss(const orc_sift *s, int oct, int lvl) { return s->gauss[oct * IMAGES + lvl]; }
const double *orc_dog(const orc_sift *s, int oct, int lvl) { return s->dog[oct * (IMAGES - 1) + lvl]; }
int orc_num_candidates(const orc_sift *s) { return s->ncand; }
const orc_candidate *orc_candidates(const orc_sift *s) { return s->cand; }
const orc_stats *orc_stats_table(const orc_sift *s) { return s->stats; }
int orc_num_keypoints(const orc_sift *s) { return s->nkp; }
const orc_keypoint *orc_keypoints(const orc_sift *s) { return s->kp; }
const double *orc_descriptors(const orc_sift *s) { return s->desc; }

/* sift_handler::get, sift.cpp:103-112. */
void orc_get(const orc_sift *s, float *out) {
    for (size_t i = 0; i < (size_t)s->nkp * 128; i++) out[i] = (float)s->desc[i];
}

/* ---- matcher oracle ---------------------------------------------------- */

typedef struct {
    const float *qf, *tf;
    const uint8_t *qb, *tb;
    int nt, dim;
    int32_t *idx1, *idx2;
    float *d1f, *d2f;
    int32_t *d1i, *d2i;
    uint8_t *accept;
} knn_ctx;

static uint8_t ratio_accept(float d1sq, float d2sq) {
    /* main.cpp:190,194: knn[0].distance < 0.7f * knn[1].distance, distances
     * are float sqrt of the squared L2 sum */
    const float ratio_thresh = 0.7f;
    return sqrtf(d1sq) < ratio_thresh * sqrtf(d2sq) ? 1 : 0;
}

static void knn_range_f32(void *vc, int tid, int b, int e) {
    (void)tid;
    knn_ctx *c = (knn_ctx *)vc;
    for (int i = b; i < e; i++) {
        const float *q = c->qf + (size_t)i * c->dim;
        double b1 = INFINITY, b2 = INFINITY;
        int i1 = -1, i2 = -1;
        for (int j = 0; j < c->nt; j++) {
            const float *t = c->tf + (size_t)j * c->dim;
            double acc = 0;
            for (int k = 0; k < c->dim; k++) {
                double df = (double)q[k] - (double)t[k];
                acc += df * df;
            }
            if (acc < b1) { b2 = b1; i2 = i1; b1 = acc; i1 = j; }
            else if (acc < b2) { b2 = acc; i2 = j; }
        }
        c->idx1[i] = i1; c->idx2[i] = i2;
        c->d1f[i] = (float)b1; c->d2f[i] = (float)b2;
        c->accept[i] = (i2 >= 0) ? ratio_accept((float)b1, (float)b2) : 0;
    }
}

static void knn_range_u8(void *vc, int tid, int b, int e) {
    (void)tid;
    knn_ctx *c = (knn_ctx *)vc;
    for (int i = b; i < e; i++) {
        const uint8_t *q = c->qb + (size_t)i * c->dim;
        int64_t b1 = INT64_MAX, b2 = INT64_MAX;
        int i1 = -1, i2 = -1;
        for (int j = 0; j < c->nt; j++) {
            const uint8_t *t = c->tb + (size_t)j * c->dim;
            int32_t acc = 0;
            for (int k = 0; k < c->dim; k++) {
                int df = (int)q[k] - (int)t[k];
                acc += df * df;
            }
            if (acc < b1) { b2 = b1; i2 = i1; b1 = acc; i1 = j; }
            else if (acc < b2) { b2 = acc; i2 = j; }
        }
        c->idx1[i] = i1; c->idx2[i] = i2;
        c->d1i[i] = i1 >= 0 ? (int32_t)b1 : INT32_MAX;
        c->d2i[i] = i2 >= 0 ? (int32_t)b2 : INT32_MAX;
        c->accept[i] = (i2 >= 0) ? ratio_accept((float)b1, (float)b2) : 0;
    }
}

void orc_knn2_f32(const float *q, int nq, const float *t, int nt, int dim, int32_t *idx1, float *d1sq, int32_t *idx2,
                  float *d2sq, uint8_t *accept, int threads) {
    knn_ctx c;
    memset(&c, 0, sizeof c);
    c.qf = q; c.tf = t; c.nt = nt; c.dim = dim;
    c.idx1 = idx1; c.idx2 = idx2; c.d1f = d1sq; c.d2f = d2sq; c.accept = accept;
    parallel_for(0, nq, threads, knn_range_f32, &c);
}

void orc_knn2_u8(const uint8_t *q, int nq, const uint8_t *t, int nt, int dim, int32_t *idx1, int32_t *d1sq,
                 int32_t *idx2, int32_t *d2sq, uint8_t *accept, int threads) {
    knn_ctx c;
    memset(&c, 0, sizeof c);
    c.qb = q; c.tb = t; c.nt = nt; c.dim = dim;
    c.idx1 = idx1; c.idx2 = idx2; c.d1i = d1sq; c.d2i = d2sq; c.accept = accept;
    parallel_for(0, nq, threads, knn_range_u8, &c);
}

/* main.cpp:77-91 */
int orc_min_dist_filter(const float *dist, int n, int32_t *good_idx) {
    double min_dist = 100;
    for (int i = 0; i < n; i++) {
        double d = dist[i];
        if (d < min_dist) min_dist = d;
    }
    int g = 0;
    for (int i = 0; i < n; i++)
        if (dist[i] < 3 * min_dist) good_idx[g++] = i;
    return g;
}
