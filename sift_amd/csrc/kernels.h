// Host-side launchers of the HIP kernels (internal to the library).
#pragma once

#include "common.h"

namespace sa {

// kernels_pyramid.hip
int blur_radius_supported(int r);
int pyramid_init_attributes();
void launch_base(hipStream_t st, const uint8_t *frames, int W, int H, int nframes, const Pyramid &p, const Taps &taps,
                 int ntaps);
void launch_blur(hipStream_t st, const Pyramid &p, int oct, int lvl, int nframes, const Taps &taps, int ntaps);
// Octaves of at most this many pixels are blurred, all levels and all such octaves, by one launch with a workgroup
// per frame (two planes and a quarter in LDS).
constexpr int SMALL_OCTAVE_PIXELS = 8192;
// First octave from which on every octave is that small (p.octaves if none is).
// Levels 1 and 2 of an octave in one launch (the first level is not read back from memory), for octaves large enough
// for it (blur_levels_1_2_pair) and the 11- and 13-tap filters.
bool blur_levels_1_2_pair(const Pyramid &p, int oct, const int *lvl_ntaps);
void launch_blur_levels_1_2(hipStream_t st, const Pyramid &p, int oct, int nframes, const Taps *lvl_taps);
int first_small_octave(const Pyramid &p);
// Levels 1..5 of the octaves o_first.. of every frame; level 0 of octave o_first must be in place.  False (and no
// launch) if the tap counts are not the five the kernel is compiled for.
bool launch_small_octaves(hipStream_t st, const Pyramid &p, int o_first, int nframes, const Taps *lvl_taps,
                          const int *lvl_ntaps);
void launch_bgr_to_gray(hipStream_t st, const uint8_t *bgr, uint8_t *gray, size_t npix);

// kernels_extrema.hip
void launch_extrema_gradients(hipStream_t st, const Pyramid &p, int oct, int nframes, Candidate *cand, int cand_cap,
                              FrameCounters *counters);
// The same for every octave from oct_first on, in one launch (for octaves of a tile or two each).
void launch_extrema_gradients_from(hipStream_t st, const Pyramid &p, int oct_first, int nframes, Candidate *cand,
                                   int cand_cap, FrameCounters *counters);
// `few_keypoints`: one or a few frames; the orientation windows get a workgroup each instead of a wave.
void launch_localize_orient(hipStream_t st, const Pyramid &p, int nframes, const Candidate *cand, int cand_cap,
                            DevKeypoint *localized, DevKeypoint *oriented, int kp_cap, FrameCounters *counters,
                            int32_t *n_localized, bool few_keypoints);

// kernels_keypoints.hip
// `ranks` ([nframes][kp_cap] of scratch, or null): when given, the rank sort is spread over eight times as many
// workgroups (for launches of one or a few frames).
void launch_clean(hipStream_t st, int nframes, const DevKeypoint *oriented, DevKeypoint *sorted, DevKeypoint *final_kp,
                  uint8_t *flags, int kp_cap, FrameCounters *counters, int32_t *ranks);

// kernels_descriptor.hip
// `launch_pixels` = frames x width x height of the launch: up to 1 M pixels keypoints get eight waves each, up to 4 M
// four, beyond that two.
void launch_descriptors(hipStream_t st, const Pyramid &p, int nframes, const DevKeypoint *kps, int kp_cap,
                        const FrameCounters *counters, int32_t *cursors, uint8_t *desc, int64_t launch_pixels);

// Per frame {contributing samples, LDS atomics} of the descriptor stage for the keypoints of the last run (`cursors` and
// `work` must be zero on entry).
void launch_descriptor_work(hipStream_t st, const Pyramid &p, int nframes, const DevKeypoint *kps, int kp_cap,
                            const FrameCounters *counters, int32_t *cursors, unsigned long long *work);

// kernels_knn.hip
size_t knn_scratch_bytes(int nq, int nt);
// Size of the file-scope device array behind sift_knn2_device_noalloc.  Partial results need at most
// 8,355,840 bytes whatever nq is (at most 522,240 split x query records of 16 bytes), train biases 4 bytes per row.
constexpr size_t KNN_STATIC_SCRATCH_BYTES = (size_t)16 << 20;
hipError_t knn_static_scratch(int device, void **ptr);
void launch_knn2(hipStream_t st, const uint8_t *q, int nq, const uint8_t *t, int nt, int train_base, sift_knn2 *out,
                 void *scratch);
void launch_knn2_merge(hipStream_t st, const sift_knn2 *parts, int nparts, int nq, sift_knn2 *out, uint8_t *accept);

// homography.hip
constexpr int HOMOGRAPHY_CHUNK = 1024;       // RANSAC iterations evaluated per launch
constexpr int HOMOGRAPHY_FIRST_CHUNK = 256;  // ... in the first launch (the budget usually collapses within it)
size_t homography_workspace_bytes(int n, int chunk);
hipError_t find_homography_ransac(hipStream_t st, void *workspace, const float *src_xy, const float *dst_xy, int n,
                                  double threshold, int max_iters, double confidence, double H[9], uint8_t *mask,
                                  sift_homography_info *info);

// kernels_stitch.hip
bool invert3x3(const double S[9], double D[9]);
void launch_warp_perspective(hipStream_t st, const uint8_t *src, int sw, int sh, const double Minv[9], int dw, int dh, uint8_t *dst);
void launch_fill_and_blend(hipStream_t st, uint8_t *canvas, int cw, const uint8_t *img, int iw, int ih);
void launch_column_grey_sums(hipStream_t st, const uint8_t *canvas, int cw, int ch, int *sums);

}  // namespace sa
