// Drop-in for the reference's src/sift.hpp (kannavkm/SIFT, sift.hpp:20-118):
// the same class, constructor, methods and public data members, with the work
// done on an MI355X through the C ABI of include/sift_amd.h.
//
//   sift::sift_handler s(name, std::move(bgr_or_gray_mat));   // sift.hpp:23
//   s.exec();                                                 // sift.hpp:25
//   cv::Mat descriptors = s.get();                            // sift.hpp:27, N x 128 CV_32FC1
//   s.keypoints;                                              // sift.hpp:93
//
// Differences from the reference, all outside the numeric results:
//   * get() on zero keypoints returns an empty 0 x 128 matrix instead of
//     indexing descriptors[0] (sift.cpp:105);
//   * "<name>_keypoints.png" (sift.cpp:89-90) is written only on request
//     (sift::write_keypoint_images(true), or SIFT_AMD_KEYPOINT_PNG=1 in the
//     environment): encoding a PNG costs ten times the extraction of the frame.
//     Its circles are not anti-aliased and its colours are not OpenCV's.
//     "<name>.csv" (sift.cpp:114-121) and the stderr lines (sift.cpp:37-38, 69,
//     93, 96 and the TIMEIT lines) are always produced;
//   * errors of the GPU library surface as std::runtime_error.
#ifndef SIFT_H
#define SIFT_H

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#define SIFT_AMD_HAVE_OPENCV 1
#else
#include "cv_compat.hpp"
#define SIFT_AMD_HAVE_OPENCV 0
#endif

#include <chrono>
#include <cstdint>
#include <iostream>
#include <string>
#include <vector>

// The reference's header also hands its users a stopwatch macro (sift.hpp:11-18): TIMEIT(f) calls f() and
// reports "elapsed time: f: <seconds>s" on stderr.
#ifndef TIMEIT
#define TIMEIT(f)                                                                                          \
    do {                                                                                                   \
        const auto timeit_begin_ = std::chrono::steady_clock::now();                                       \
        f();                                                                                               \
        const std::chrono::duration<double> timeit_span_ = std::chrono::steady_clock::now() - timeit_begin_; \
        std::cerr << "elapsed time: " << #f << ": " << timeit_span_.count() << "s\n";                      \
    } while (0)
#endif

namespace sift {

class sift_handler {
   public:
    explicit sift_handler(std::string _name, cv::Mat &&_base);

    void exec();

    [[nodiscard]] cv::Mat get() const;

    ~sift_handler();

    sift_handler(const sift_handler &) = delete;
    sift_handler &operator=(const sift_handler &) = delete;

   public:
    cv::Mat base, onex;
    int32_t octaves;
    std::string name;
    std::vector<cv::KeyPoint> keypoints;
    std::vector<std::vector<double>> descriptors;

   private:
    void dump_keypoints();

    std::vector<unsigned char> gray_;  // 8-bit gray input, row-major
    int width_ = 0, height_ = 0;
};

// Whether exec() writes "<name>_keypoints.png" (sift.cpp:89-90).  Off unless SIFT_AMD_KEYPOINT_PNG=1 is set when
// the first handler runs; returns the previous setting.
bool write_keypoint_images(bool on);

// Matcher used in place of cv::FlannBasedMatcher (main.cpp:48-51, 75): exact
// top-2 nearest neighbours on the GPU.  query/train are N x 128 CV_32FC1
// matrices as returned by sift_handler::get().
struct match_result {
    int queryIdx = -1;
    int trainIdx = -1;   // nearest train descriptor
    float distance = 0;  // L2 distance to it (DMatch::distance)
    int trainIdx2 = -1;  // second nearest
    float distance2 = 0;
    bool ratio_ok = false;  // distance < 0.7f * distance2 (main.cpp:190-194)
};
std::vector<match_result> knn_match(const cv::Mat &query, const cv::Mat &train);
// main.cpp:77-91: min_dist starts at 100; good <=> distance < 3 * min_dist.
std::vector<match_result> good_matches(const std::vector<match_result> &matches);

// cv::findHomography(obj, scene, RANSAC, ransacReprojThreshold, mask) as main.cpp:105
// calls it: the 3x3 matrix (row-major) that maps obj to scene and the inlier mask.
// `found` is false where OpenCV returns an empty matrix (fewer than 4 points, or
// no usable model); H is then all zeros.
struct homography_result {
    bool found = false;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<unsigned char> mask;  // one byte per correspondence, 1 = inlier
    int iterations = 0;               // RANSAC iterations
    int inliers = 0;
};
homography_result find_homography(const std::vector<cv::Point2f> &obj, const std::vector<cv::Point2f> &scene,
                                  double ransacReprojThreshold = 3, int maxIters = 2000, double confidence = 0.995);

// The composition step of main.cpp:115-145 for 8-bit BGR images: `ref` warped by H (3x3 row-major, as
// find_homography returns it) into a canvas of ref.cols + img.cols columns, holes filled from `img`, the 0.7 / 0.3
// blend over img's rectangle, cropped by the number of all-black columns.  img must not be taller than ref (the
// reference throws there).
cv::Mat stitch_pair(const cv::Mat &ref, const cv::Mat &img, const double H[9]);
// cv::warpPerspective(src, dst, H, dsize) with the defaults main.cpp:115 uses (bilinear, black border).
cv::Mat warp_perspective(const cv::Mat &src, const double H[9], cv::Size dsize);

}  // namespace sift

#endif
