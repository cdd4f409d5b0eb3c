/*
 * sift_oracle.h -- CPU restatement of kannavkm/SIFT's feature engine.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker that the HIP engine
 * is compared against; nothing in the product path may link, load or call it.
 *
 * It follows /root/reference/src/sift.cpp function by function (citations at
 * each definition in sift_oracle.c) in double precision, with the OpenCV
 * primitives the reference calls (cvtColor, resize, GaussianBlur, subtract,
 * solve, KeyPoint narrowing) restated from OpenCV 4.x's published scalar
 * algorithms.  OpenCV is a third-party dependency of the reference that is not
 * vendored (CMakeLists.txt:9, no version pin), so those restatements are pinned
 * only by the reference's single published known answer
 * (README.md:353-354: 512x512 Lenna -> 1112 keypoints): "parity unpinned"
 * beyond that count.
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order and widths as cv::KeyPoint (pt.x, pt.y, size, angle,
 * response, octave, class_id): 28 bytes. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

/* Candidate = a pixel that passed is_pixel_extremum (sift.cpp:264-281). */
typedef struct {
    int32_t oct, img, i, j;
} orc_candidate;

/* Outcome counters of localize_extrema (sift.cpp:312-387), per (oct, img 1..3). */
typedef struct {
    int64_t candidates;
    int64_t out_of_bounds;   /* sift.cpp:347-350 */
    int64_t no_convergence;  /* sift.cpp:356-358 */
    int64_t low_contrast;    /* sift.cpp:363 */
    int64_t det_nonpositive; /* sift.cpp:367-369 */
    int64_t edge_like;       /* sift.cpp:374 */
    int64_t singular;        /* sift.cpp:333-335 */
    int64_t localized;       /* returned a layer index */
    int64_t oriented;        /* keypoints pushed by get_keypoint_orientations */
} orc_stats;

typedef struct orc_sift orc_sift;

/* cvtColor(BGR2GRAY) on 8-bit data, sift.cpp:29. */
void orc_bgr_to_gray(const uint8_t *bgr, int w, int h, uint8_t *gray);

/* Constructor, sift.cpp:26-46, starting after cvtColor (gray 8-bit input). */
orc_sift *orc_create_gray(const uint8_t *gray, int w, int h, int stride);
void orc_destroy(orc_sift *s);

/* The five TIMEIT stages of exec(), sift.cpp:71,72,86,87,92.
 * scan_threads splits rows (sift.cpp:188) / keypoints (sift.cpp:452) the way
 * cv::parallel_for_ does; blur_threads > 1 additionally splits blur rows (the
 * reference's GaussianBlur on CV_64F is serial). */
void orc_gen_gaussian_images(orc_sift *s, int blur_threads);
void orc_gen_dog_images(orc_sift *s);
void orc_gen_scale_space_extrema(orc_sift *s, int scan_threads);
void orc_clean_keypoints(orc_sift *s);
void orc_get_descriptors(orc_sift *s, int scan_threads);
/* All five in order, recording wall seconds per stage in timings[5]. */
void orc_exec(orc_sift *s, int scan_threads, int blur_threads, double *timings);

int orc_octaves(const orc_sift *s);
void orc_level_dims(const orc_sift *s, int oct, int *w, int *h);
const double *orc_onex(const orc_sift *s);
const double *orc_base(const orc_sift *s);
const double *orc_gauss(const orc_sift *s, int oct, int lvl); /* lvl 0..5 */
const double *orc_dog(const orc_sift *s, int oct, int lvl);   /* lvl 0..4 */

int orc_num_candidates(const orc_sift *s);
const orc_candidate *orc_candidates(const orc_sift *s); /* sorted (oct,img,i,j) */
/* stats[oct*3 + (img-1)] */
const orc_stats *orc_stats_table(const orc_sift *s);

int orc_num_keypoints(const orc_sift *s);
const orc_keypoint *orc_keypoints(const orc_sift *s);
const double *orc_descriptors(const orc_sift *s); /* n x 128, row-major */
/* get(), sift.cpp:103-112: double -> float copy. */
void orc_get(const orc_sift *s, float *out);

/* Gaussian kernel exactly as the blur uses it (for known-answer tests).
 * Returns the tap count; coeffs must hold at least 64 doubles. */
int orc_gaussian_kernel(double sigma, double *coeffs);

/*
 * Exact brute-force top-2 nearest neighbours under squared L2 distance plus
 * Lowe's ratio test (the k=2 / 0.7f variant at main.cpp:186-198, 254-267).
 * Ties are ordered by (distance, train index).  q and t are row-major
 * [n][dim] float descriptors as produced by get().  accept[i] = 1 when
 * sqrtf(d1) < 0.7f * sqrtf(d2).  With nt < 2 the missing neighbour has index
 * -1, distance +inf and accept = 0.
 */
void orc_knn2_f32(const float *q, int nq, const float *t, int nt, int dim,
                  int32_t *idx1, float *d1sq, int32_t *idx2, float *d2sq,
                  uint8_t *accept, int threads);
/* Same on byte descriptors (integer squared distances). */
void orc_knn2_u8(const uint8_t *q, int nq, const uint8_t *t, int nt, int dim,
                 int32_t *idx1, int32_t *d1sq, int32_t *idx2, int32_t *d2sq,
                 uint8_t *accept, int threads);

/*
 * The live matcher filter of main.cpp:77-91 applied to 1-NN distances
 * (dist[i] = sqrtf(d1sq[i])): min_dist starts at 100, good <=> dist < 3*min_dist.
 * Returns the number of good matches and writes their query indices.
 */
int orc_min_dist_filter(const float *dist, int n, int32_t *good_idx);

#ifdef __cplusplus
}
#endif
#endif
