/*
 * sift_amd.h -- C ABI of the MI355X (gfx950) SIFT engine.
 *
 * The reference (kannavkm/SIFT) has no FFI layer; its seam is the C++ class
 * sift::sift_handler (src/sift.hpp:20-118) used by src/main.cpp.  This header is
 * the boundary a binding of that class talks to: plain pointers and sizes, no
 * C++/torch types.  Each entry point names the reference interface it
 * replaces.  All functions return 0 on success and a negative SIFT_ERR_* code
 * on failure; sift_last_error() gives the message of the calling thread's last
 * failure.  Nothing throws.
 *
 * Threading: a context (and every batch created from it) is bound to one GPU
 * and must be used from one thread at a time; distinct contexts are
 * independent.
 */
#ifndef SIFT_AMD_H
#define SIFT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT_OK 0
#define SIFT_ERR_INVALID (-1)  /* bad argument */
#define SIFT_ERR_HIP (-2)      /* a HIP runtime call failed */
#define SIFT_ERR_OVERFLOW (-3) /* a per-frame candidate/keypoint list exceeded its capacity */
#define SIFT_ERR_NOT_INTEGRAL (-4) /* float descriptors given to the matcher are not integers in [0,127] */

#define SIFT_DESCRIPTOR_DIM 128

/* Same field order and widths as cv::KeyPoint as the reference fills it
 * (sift.cpp:375-382, :443): pt.x, pt.y, size, angle, response are float32,
 * octave = oct + 256*layer + 65536*round((offset+0.5)*255), class_id = -1. */
typedef struct sift_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} sift_keypoint;

typedef struct sift_ctx sift_ctx;     /* one per GPU */
typedef struct sift_batch sift_batch; /* pyramid workspace for up to max_frames same-sized frames */

/* Per-frame outcome of sift_batch_run. */
typedef struct sift_frame_counts {
    int32_t candidates; /* pixels passing is_pixel_extremum (sift.cpp:264-281) */
    int32_t oriented;   /* keypoints before clean_keypoints (sift.cpp:183-197) */
    int32_t keypoints;  /* keypoints after clean_keypoints (sift.cpp:202-225) */
    int32_t overflow;   /* non-zero when a list exceeded its capacity; results of this frame are incomplete */
} sift_frame_counts;

/* Result of the single-image convenience call; buffers are owned by the
 * library and released by sift_result_free. */
typedef struct sift_result {
    int32_t n;                /* keypoints.size() */
    int32_t octaves;          /* sift_handler::octaves (sift.hpp:91) */
    sift_keypoint *keypoints; /* sift_handler::keypoints (sift.hpp:93), sorted as clean_keypoints leaves them */
    double *descriptors;      /* sift_handler::descriptors (sift.hpp:94): n x 128 integer-valued doubles */
    sift_frame_counts counts;
} sift_result;

const char *sift_last_error(void);
int sift_device_count(int *count);

/* ---- context ----------------------------------------------------------- */
int sift_ctx_create(int device, sift_ctx **out);
int sift_ctx_destroy(sift_ctx *ctx);
/* The HIP stream (hipStream_t) every call of this context enqueues on. */
int sift_ctx_stream(sift_ctx *ctx, void **stream);
int sift_ctx_synchronize(sift_ctx *ctx);
/* Initial capacity (candidates and keypoints per frame) of the workspaces behind sift_extract_* and
 * sift_extract_batch_u8; 0 restores the default, max(8192, pixels/16).  The reference's vectors (sift.hpp:93-94) grow without bound: when a frame
 * exceeds the capacity, sift_extract_* runs it again in a larger workspace, so this is a memory knob only. */
int sift_ctx_set_single_image_capacity(sift_ctx *ctx, int max_keypoints);
/* Diagnostic (no counterpart in the reference): how many times the extraction pipeline has been launched on this
 * context, whatever entry point launched it; lets a test see how sift_extract_batch_u8 cut a run into launches. */
int sift_ctx_pipeline_runs(sift_ctx *ctx, int64_t *runs);

/* ---- geometry helpers (host only) ---------------------------------------- */
/* octaves = round(log2(min(w,h))) on the input size (sift.cpp:36). */
int sift_num_octaves(int w, int h);
/* Size of pyramid octave `oct` for a w x h input: 2w x 2h at octave 0
 * (sift.cpp:40), then nearest-neighbour halving with cvRound sizes
 * (sift.cpp:153). */
int sift_octave_dims(int w, int h, int oct, int *ow, int *oh);

/* ---- single image: constructor + exec() + get() ------------------------- */
/* Replaces sift_handler(name, Mat&&) + exec() (sift.hpp:23,25) for one 8-bit
 * image held in host memory.  sift_extract_bgr8 starts at cvtColor(BGR2GRAY)
 * (sift.cpp:29); sift_extract_gray8 starts after it.  stride is in bytes. */
int sift_extract_gray8(sift_ctx *ctx, const uint8_t *gray, int w, int h, size_t stride, sift_result *out);
int sift_extract_bgr8(sift_ctx *ctx, const uint8_t *bgr, int w, int h, size_t stride, sift_result *out);
/* sift_extract_gray8 that also reports the device time of the five stages the reference times with TIMEIT in exec()
 * (sift.cpp:71,72,86,87,92), in milliseconds: gen_gaussian_images, gen_dog_images (always 0: the differences are never
 * stored), gen_scale_space_extrema, clean_keypoints, get_descriptors.  The schedule is that of sift_extract_gray8; the
 * times come from five events at the stage boundaries of the main stream.  Frames of up to 4 M pixels take a schedule
 * in which blur levels 4-5 of the large octaves run on side streams beside the extrema scans: the boundary between the
 * first and the third figure is then the end of levels 1-3 of the last octave, i.e. gen_gaussian_images leaves out the
 * levels 4-5 that are still running and gen_scale_space_extrema takes them in.  Their sum is exact, the split between
 * the two is approximate; the other boundaries are as the reference's. */
int sift_extract_gray8_stages(sift_ctx *ctx, const uint8_t *gray, int w, int h, size_t stride, sift_result *out,
                              double stage_ms[5]);
/* The base image the constructor builds (sift.cpp:40-45): the gray frame upsampled x2 (INTER_LINEAR) and blurred
 * with sigma sqrt(1.6^2 - 4*0.5^2); base receives 2h rows of 2w doubles. */
int sift_base_image_gray8(sift_ctx *ctx, const uint8_t *gray, int w, int h, size_t stride, double *base);
int sift_result_free(sift_result *r);
/* get() (sift.hpp:27, sift.cpp:103-112): n x 128 float32 copy of the descriptors. */
int sift_result_get_f32(const sift_result *r, float *out);

/* ---- batch of frames resident in device memory --------------------------- */
/* Allocates frame, pyramid, candidate and keypoint buffers for up to
 * max_frames frames of w x h.  max_keypoints_per_frame <= 0 selects a default
 * derived from the frame size. */
int sift_batch_create(sift_ctx *ctx, int w, int h, int max_frames, int max_keypoints_per_frame, sift_batch **out);
int sift_batch_destroy(sift_batch *b);
/* Device pointer of the 8-bit gray frame buffer, [max_frames][h][w], for
 * callers that fill frames on the GPU themselves. */
int sift_batch_frames_device_ptr(sift_batch *b, void **dptr);
/* Copy one host frame into slot `frame`.  The call waits for earlier work on the context stream, copies the
 * frame into the context's page-locked staging buffer (the caller's buffer may be reused on return) and enqueues the
 * transfer on the stream. */
int sift_batch_upload_gray8(sift_batch *b, int frame, const uint8_t *gray, size_t stride);
int sift_batch_upload_bgr8(sift_batch *b, int frame, const uint8_t *bgr, size_t stride);
/* Make the next runs read their frames from a caller-owned device buffer laid
 * out [n][h][w] bytes instead of the batch's own frame buffer (NULL restores
 * the internal buffer).  The buffer must stay valid until the run completes. */
int sift_batch_set_external_frames(sift_batch *b, const void *d_frames);
/* Enqueue the whole of exec() (sift.cpp:67-98: gen_gaussian_images,
 * gen_dog_images, gen_scale_space_extrema, clean_keypoints, get_descriptors)
 * for frames [0, n_frames) on the context stream.  Asynchronous. */
int sift_batch_run(sift_batch *b, int n_frames);
/* Wait for the stream, then report per-frame counts.  Returns
 * SIFT_ERR_OVERFLOW if any frame overflowed (counts are still filled). */
int sift_batch_counts(sift_batch *b, int n_frames, sift_frame_counts *counts);
/* Copy frame results to the host.  keypoints/desc_u8/desc_f64 may be NULL;
 * each must hold `capacity` keypoints (x128 for descriptors).  *n receives the
 * frame's keypoint count. */
int sift_batch_download(sift_batch *b, int frame, int capacity, sift_keypoint *keypoints, uint8_t *desc_u8,
                        double *desc_f64, int32_t *n);
/* Device pointers of a frame's final keypoints (32-byte records whose first 28
 * bytes are a sift_keypoint) and 128-byte descriptors, valid until the next run. */
int sift_batch_result_device_ptrs(sift_batch *b, int frame, void **keypoints, void **desc_u8);

/* Measurement aid: the work the descriptor stage (sift.cpp:464-584) did for the keypoints of the last completed run,
 * counted by a pass over the same window samples: work[2*f] = samples that contribute to frame f's descriptors
 * (inside the rotated 4x4 window, non-zero gradient), work[2*f+1] = histogram updates they make (two orientation bins
 * for each of up to four cells).  Synchronous. */
int sift_batch_descriptor_work(sift_batch *b, int n_frames, int64_t *work);

/* Parity/debug accessors (host copies of intermediate images and lists). */
#define SIFT_LEVEL_GAUSS 0 /* gauss_images[oct][lvl], lvl 0..5 (sift.hpp:115) */
#define SIFT_LEVEL_DOG 1   /* images[oct][lvl], lvl 0..4 (sift.hpp:114) */
/* Per-pixel gradient of gauss_images[oct][lvl], lvl 1..3, as the orientation and descriptor windows evaluate it per
 * sample (sift.cpp:410-415, :507-510): magnitude sqrt(dx^2 + dy^2) and direction atan2(dy, dx)*180/pi with
 * dx = L(y,x+1) - L(y,x-1), dy = L(y-1,x) - L(y+1,x); zero on the one-pixel image border, which no window samples. */
#define SIFT_LEVEL_GRAD_MAG 2
#define SIFT_LEVEL_GRAD_DIR 3
int sift_batch_get_level(sift_batch *b, int frame, int oct, int lvl, int kind, double *out);
/* Candidates as (oct, img, i, j) int32 quadruples in unspecified order. */
int sift_batch_get_candidates(sift_batch *b, int frame, int capacity, int32_t *quads, int32_t *n);
/* Keypoints before clean_keypoints, unspecified order. */
int sift_batch_get_oriented(sift_batch *b, int frame, int capacity, sift_keypoint *kps, int32_t *n);

/* Per-kernel device time of one run, measured with HIP events on the context
 * stream around every launch (serialises the run; use for profiling only). */
#define SIFT_STAGE_BASE 0       /* gray -> 2x upsample -> base blur (sift.cpp:40-42) */
#define SIFT_STAGE_BLUR_L1 1    /* GaussianBlur level 1..5 + subtract (+ next-octave resize at level 3), */
#define SIFT_STAGE_BLUR_L2 2    /* one stage per level = per tap count 11, 13, 17, 21, 27                */
#define SIFT_STAGE_BLUR_L3 3    /* (sift.cpp:150-153, :173)                                              */
#define SIFT_STAGE_BLUR_L4 4
#define SIFT_STAGE_BLUR_L5 5
#define SIFT_STAGE_EXTREMA 6    /* 3x3x3 scan (sift.cpp:227-281) */
#define SIFT_STAGE_LOCALIZE 7   /* localize_extrema + get_keypoint_orientations (sift.cpp:312-448) */
#define SIFT_STAGE_CLEAN 8      /* clean_keypoints (sift.cpp:202-225) */
#define SIFT_STAGE_DESCRIPTOR 9 /* get_descriptors (sift.cpp:450-584) */
#define SIFT_STAGE_GRADIENT 10  /* per-pixel gradient magnitude/direction of levels 1..3, shared by the orientation
                                   and descriptor windows (sift.cpp:410-415, :507-510) */
#define SIFT_STAGE_BLUR_L12 11 /* levels 1 and 2 of a large octave in one launch (the batch schedule) */
#define SIFT_STAGE_COUNT 12
typedef struct sift_stage_times {
    double ms[SIFT_STAGE_COUNT];    /* summed kernel time per stage */
    int32_t launches[SIFT_STAGE_COUNT];
} sift_stage_times;
int sift_batch_run_profiled(sift_batch *b, int n_frames, sift_stage_times *times);

/* ---- matcher: exact brute-force top-2 + ratio test ------------------------ */
/* Replaces FlannBasedMatcher::match / knnMatch(k=2) at main.cpp:48-51,75 and
 * the ratio filter at main.cpp:186-198 with an exact search: for every query
 * the two nearest train descriptors under squared L2, ordered by (distance,
 * train index). */
typedef struct sift_knn2 {
    int32_t d1; /* squared distance to the nearest train descriptor (INT32_MAX if none) */
    int32_t i1; /* its index (-1 if none) */
    int32_t d2; /* second nearest */
    int32_t i2;
} sift_knn2;

/* Host buffers, byte descriptors with values in [0,127]; [n][128] row-major.
 * accept[i] = sqrtf(d1) < 0.7f*sqrtf(d2) (main.cpp:190,194); accept may be NULL. */
int sift_knn2_u8(sift_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, sift_knn2 *out, uint8_t *accept);
/* Same for float descriptors as get() returns them; values must be integers in
 * [0,127] (they always are for this engine: sift.cpp:566-573), otherwise
 * SIFT_ERR_NOT_INTEGRAL. */
int sift_knn2_f32(sift_ctx *ctx, const float *q, int nq, const float *t, int nt, sift_knn2 *out, uint8_t *accept);
/* Device-resident operands (asynchronous on the context stream).  d_q and d_t
 * are device pointers to [n][128] bytes; d_out receives nq sift_knn2 records
 * whose indices are offset by train_index_base (for a train shard). */
int sift_knn2_device(sift_ctx *ctx, const void *d_q, int nq, const void *d_t, int nt, int train_index_base,
                     void *d_out);
/* sift_knn2_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same results;
 * the train biases and per-split partial results live in a 16 MiB array that is part of the library's code object
 * (every GPU the library is used on holds its own copy until the process ends) instead of a buffer the context
 * allocates on demand, and the call makes no hipMalloc/hipFree.  Limit: nt <= SIFT_KNN2_NOALLOC_MAX_TRAIN train rows
 * per call, any nq (shard a larger train set with train_index_base and merge with sift_knn2_merge_device); beyond it
 * the call fails with SIFT_ERR_INVALID and enqueues nothing.  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (two contexts, two streams) race; calls on one context are ordered by its stream. */
#define SIFT_KNN2_NOALLOC_MAX_TRAIN 2097152
int sift_knn2_device_noalloc(sift_ctx *ctx, const void *d_q, int nq, const void *d_t, int nt, int train_index_base,
                             void *d_out);
/* Merge `nparts` partial results per query (d_parts laid out [nparts][nq]) into
 * d_out[nq] ordered by (distance, index), and apply the ratio test into
 * d_accept[nq] (may be NULL).  Device pointers, asynchronous. */
int sift_knn2_merge_device(sift_ctx *ctx, const void *d_parts, int nparts, int nq, void *d_out, void *d_accept);

/* The live filter of main.cpp:77-91 on 1-NN results: min_dist starts at 100,
 * a query is good when sqrtf(d1) < 3*min_dist.  Host only.  Returns the number
 * of good matches through *n_good and their query indices in good_idx. */
int sift_match_min_dist_filter(const sift_knn2 *knn, int nq, int32_t *good_idx, int32_t *n_good);

/* ---- all GPUs of a node from one process ---------------------------------------- */
/* The reference is one C++ program (main.cpp:22-152); a group lets such a program spread the two stages that shard
 * over several GPUs without leaving the process: frames are independent (contiguous blocks of frames per GPU, no
 * collective), and the matcher's train set is cut into contiguous shards whose per-GPU top-2 results are exchanged
 * with RCCL and merged.  A group borrows its contexts, normally one per GPU; they must outlive it.  (Several
 * contexts of one GPU may share a group: sift_extract_batch_u8 then splits the frames between them, and
 * sift_knn2_sharded moves the records with device-to-device copies ordered by events instead of RCCL, which has
 * one rank per GPU.)
 * Calls on a group must not overlap with other calls on its contexts. */
typedef struct sift_group sift_group;
int sift_group_create(sift_ctx *const *ctxs, int ngpu, sift_group **out);
int sift_group_destroy(sift_group *g);
int sift_group_size(const sift_group *g);

/* Host only: the contiguous near-equal split used for frames and train rows, part r = [bounds[r], bounds[r+1]),
 * bounds[r] = n*r/parts; bounds holds parts+1 entries. */
int sift_shard_bounds(int64_t n, int parts, int64_t *bounds);
/* Host only: the query slices of the sharded matcher's exchange.  Every part sends `slice_len` records (the last
 * slice is padded) of its partial results to each peer; part `rank` merges and owns queries [own_lo, own_hi). */
int sift_knn2_exchange_plan(int nq, int parts, int rank, int32_t *slice_len, int32_t *own_lo, int32_t *own_hi);

/* exec() (sift.hpp:25) over n same-sized 8-bit gray frames held in host memory, frame i at
 * frames + i*frame_stride with row_stride bytes per row: frames are dealt to the GPUs of the group in contiguous
 * blocks, and every GPU runs its block frames_per_launch frames at a time through two workspaces, so that
 * the upload of the next chunk and the download of the previous one overlap the kernels.  frames_per_launch <= 0
 * leaves the launch sizes to the library: at most -frames_per_launch frames (16 for 0), with a quarter-size launch
 * first and last (the GPU starts sooner and the last download, which nothing overlaps, is shorter).  results[i] receives
 * frame i's keypoints (clean_keypoints order) and 128-byte descriptors in library-owned buffers released by
 * sift_frame_results_free.  A frame that exceeds the workspace's list capacity is run again on its own in a larger
 * one, as sift_extract_gray8 does. */
typedef struct sift_frame_result {
    int32_t n;
    sift_keypoint *keypoints;
    uint8_t *descriptors; /* n x 128, values 0..255 (at most 102 in practice, sift.cpp:566-573) */
    sift_frame_counts counts;
} sift_frame_result;
int sift_extract_batch_u8(sift_group *g, const uint8_t *frames, int n, int w, int h, size_t row_stride, size_t frame_stride,
                          int frames_per_launch, sift_frame_result *results);
int sift_frame_results_free(sift_frame_result *results, int n);

/* sift_knn2_u8 with the train set sharded over the GPUs of the group (main.cpp:75 at a size one GPU should not
 * hold): GPU r searches train rows [bounds[r], bounds[r+1]) for all queries, the partial top-2 records (16 bytes per
 * query) of query slice s travel to GPU s (grouped ncclSend/ncclRecv; copies when members share a GPU), which
 * merges them under (distance, index) order and applies the ratio test.  Results are identical to sift_knn2_u8
 * whatever the number of members. */
int sift_knn2_sharded(sift_group *g, const uint8_t *q, int nq, const uint8_t *t, int nt, sift_knn2 *out, uint8_t *accept);

/* ---- homography from matches ------------------------------------------------ */
/* Replaces cv::findHomography(obj, scene, RANSAC, 3, mask) at main.cpp:105 (OpenCV
 * calib3d: RANSAC over 4-point normalised-DLT models with cv::RNG((uint64)-1),
 * a least-squares fit over the inliers, 10 Levenberg-Marquardt iterations).
 * The 4-point models and their inlier counts are evaluated on the GPU for a
 * whole chunk of iterations at once; the outcome is the sequential loop's. */
typedef struct sift_homography_info {
    int32_t found;             /* 0: no model (fewer than 4 points, or no usable subset); H is all zeros */
    int32_t iterations;        /* RANSAC iterations the sequential loop would have executed */
    int32_t best_iteration;    /* the iteration whose model was kept */
    int32_t inliers;           /* non-zeros of mask */
    int32_t lm_iterations;     /* Levenberg-Marquardt iterations of the final polish */
    int32_t hypotheses_scored; /* 4-point models evaluated on the GPU (>= iterations) */
} sift_homography_info;
/* src_xy, dst_xy: n interleaved (x, y) float pairs on the host (the Point2f vectors
 * of main.cpp:92-101).  H: 3x3 row-major, maps src to dst.  mask (may be NULL): n
 * bytes, 1 for inliers.  OpenCV's defaults are max_iters 2000, confidence 0.995.
 * SIFT_ERR_INVALID unless 0 < confidence < 1, threshold > 0, n >= 0. */
int sift_find_homography(sift_ctx *ctx, const float *src_xy, const float *dst_xy, int n, double reproj_threshold,
                         int max_iters, double confidence, double H[9], uint8_t *mask, sift_homography_info *info);

/* ---- panorama composition ------------------------------------------------------ */
/* Images are 8-bit BGR on the host, `stride` bytes per row; outputs are tightly packed.
 *
 * cv::warpPerspective(src, dst, H, Size(dw, dh)) as main.cpp:115 calls it: INTER_LINEAR,
 * BORDER_CONSTANT black, H maps src to dst.  SIFT_ERR_INVALID when H is singular. */
int sift_warp_perspective_bgr8(sift_ctx *ctx, const uint8_t *src, int sw, int sh, size_t stride, const double H[9], int dw,
                               int dh, uint8_t *dst);
/* main.cpp:115-145: warp `ref` by H into a canvas of ref_w + img_w columns and ref_h rows, take `img` where a
 * channel of the canvas is zero, blend 0.7 img + 0.3 canvas over img's rectangle, and drop as many columns from
 * the right as there are all-black columns.  out must hold ref_h * (ref_w + img_w) * 3 bytes; its rows are packed
 * at the cropped width *out_w.  SIFT_ERR_INVALID when img is taller than ref (the reference throws) or H is
 * singular. */
int sift_stitch_pair_bgr8(sift_ctx *ctx, const uint8_t *ref, int ref_w, int ref_h, size_t ref_stride, const uint8_t *img,
                          int img_w, int img_h, size_t img_stride, const double H[9], uint8_t *out, int *out_w);

/* ---- synthetic inputs (host) ---------------------------------------------- */
/* Dead-leaves frame generator: splitmix64 seeded with
 * seed*0x9E3779B97F4A7C15 + frame_index; w*h/128 discs painted over a 128-gray
 * canvas in draw order.  Integer-only, so identical everywhere. */
int sift_synth_dead_leaves(uint64_t seed, uint64_t frame_index, int w, int h, uint8_t *gray);
/* SIFT-like byte descriptors in [0,102]: n rows of 128. */
int sift_synth_descriptors(uint64_t seed, int64_t n, uint8_t *out);
/* Queries: for even i a noisy copy (noise in [-2,2], clamped to [0,102]) of a
 * pseudo-random train row, for odd i an independent descriptor. */
int sift_synth_queries(uint64_t seed, const uint8_t *train, int64_t nt, int64_t nq, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
