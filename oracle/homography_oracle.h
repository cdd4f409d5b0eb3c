/* CPU restatement of findHomography(RANSAC) (main.cpp:105); test infrastructure only.
 * Parity unpinned: see the header of homography_oracle.c. */
#ifndef HOMOGRAPHY_ORACLE_H
#define HOMOGRAPHY_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_h_info {
    int32_t found;          /* a model with at least 4 inliers exists */
    int32_t iterations;     /* RANSAC iterations executed */
    int32_t best_iteration; /* iteration that produced the kept model */
    int32_t inliers;
    int32_t lm_iterations;
} orc_h_info;

/* first n outputs of cv::RNG(state).next() */
void orc_h_rng_sequence(uint64_t state, int n, uint32_t *out);
/* the 4-point subsets RANSAC draws in its first n_iters iterations; returns how many were produced */
int orc_h_subsets(const float *src, const float *dst, int count, int n_iters, int32_t *idx_out);
/* normalised DLT on `count` correspondences; returns 0 when degenerate (H untouched) */
int orc_h_kernel(const float *src, const float *dst, int count, double H[9]);
/* number of points with float reprojection error <= thresh^2; mask may be NULL */
int orc_h_inliers(const float *src, const float *dst, int count, const double H[9], double thresh, uint8_t *mask);
/* Levenberg-Marquardt refinement of H[0..7]; returns the iteration count */
int orc_h_refine(const float *src, const float *dst, int count, double H[9], int max_iters);
/* points are interleaved x,y pairs; returns found */
int orc_find_homography(const float *src, const float *dst, int n, double thresh, int max_iters, double confidence,
                        double H[9], uint8_t *mask, orc_h_info *info);

#ifdef __cplusplus
}
#endif
#endif
