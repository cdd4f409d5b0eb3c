// sift::sift_handler over the C ABI (see sift.hpp).  Mirrors the control flow
// and side effects of the reference's src/sift.cpp constructor, exec(), get()
// and dump_keypoints(); every numeric stage runs in libsift_amd.so.
#include "sift.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <stdexcept>

#include "sift_amd.h"

namespace sift {

namespace {

void check(int rc, const char *what) {
    if (rc != SIFT_OK) throw std::runtime_error(std::string(what) + ": " + sift_last_error());
}

// One context per process, on the device named by SIFT_AMD_DEVICE (default 0).
sift_ctx *context() {
    static sift_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = std::getenv("SIFT_AMD_DEVICE");
        check(sift_ctx_create(env ? std::atoi(env) : 0, &ctx), "sift_ctx_create");
    });
    return ctx;
}

// The pyramid workspace of the last frame size, kept between exec() calls (allocating and freeing it costs more
// than running a frame through it).  The library is used from one thread at a time, as the reference is.
// `min_capacity` (keypoints per frame, 0 = the library's default) only ever grows the workspace.
sift_batch *workspace(int w, int h, int min_capacity) {
    static sift_batch *batch = nullptr;
    static int bw = 0, bh = 0, cap = 0;
    if (!batch || bw != w || bh != h || cap < min_capacity) {
        if (batch) sift_batch_destroy(batch);
        batch = nullptr;
        check(sift_batch_create(context(), w, h, 1, min_capacity, &batch), "sift_batch_create");
        bw = w;
        bh = h;
        cap = min_capacity;
    }
    return batch;
}

}  // namespace

// sift.cpp:26-46: gray conversion, onex, octave count; the 2x upsample and base
// blur happen on the GPU inside exec().
sift_handler::sift_handler(std::string _name, cv::Mat &&_base) : name(std::move(_name)) {
    if (_base.empty() || _base.depth() != CV_8U || (_base.channels() != 3 && _base.channels() != 1))
        throw std::runtime_error("sift_handler: expected an 8-bit BGR or gray image");
    width_ = _base.cols;
    height_ = _base.rows;
    gray_.resize((size_t)width_ * height_);
    for (int y = 0; y < height_; y++) {
        const unsigned char *row = _base.ptr<unsigned char>(y);
        unsigned char *dst = gray_.data() + (size_t)y * width_;
        if (_base.channels() == 1) {
            std::memcpy(dst, row, (size_t)width_);
        } else {
            // cvtColor(BGR2GRAY), sift.cpp:29: OpenCV 4.x 15-bit integer coefficients
            for (int x = 0; x < width_; x++)
                dst[x] = (unsigned char)((row[3 * x] * 3735 + row[3 * x + 1] * 19235 + row[3 * x + 2] * 9798 + (1 << 14)) >> 15);
        }
    }
    _base.release();  // sift.cpp:32
    onex = cv::Mat(height_, width_, CV_64FC1);  // sift.cpp:30, :33
    for (int y = 0; y < height_; y++)
        for (int x = 0; x < width_; x++) onex.at<double>(y, x) = gray_[(size_t)y * width_ + x];
    octaves = sift_num_octaves(width_, height_);  // sift.cpp:36
    std::cerr << "Octaves: " << octaves << std::endl;
    std::cerr << "Size: " << cv::Size(width_, height_) << std::endl;
}

sift_handler::~sift_handler() {
    base.release();
    onex.release();
    keypoints.clear();
    descriptors.clear();
}

// sift.cpp:67-98
void sift_handler::exec() {
    std::cerr << "Running SIFT for " << name << "\n";
    sift_batch *batch = nullptr;
    sift_stage_times t;
    sift_frame_counts counts;
    // keypoints and descriptors are unbounded vectors in the reference (sift.hpp:93-94): a frame that overflows
    // the workspace is run again in a larger one
    for (int capacity = 0, attempt = 0;; attempt++) {
        batch = workspace(width_, height_, capacity);
        check(sift_batch_upload_gray8(batch, 0, gray_.data(), (size_t)width_), "sift_batch_upload_gray8");
        check(sift_batch_run_profiled(batch, 1, &t), "sift_batch_run_profiled");
        const int rc = sift_batch_counts(batch, 1, &counts);
        if (rc == SIFT_ERR_OVERFLOW && attempt < 4) {
            capacity = 2 * std::max({counts.candidates, counts.oriented, 8192, width_ * height_ / 16});
            continue;
        }
        check(rc, "sift_batch_counts");
        break;
    }
    {
        // the reference's five TIMEIT stages (sift.cpp:71,72,86,87,92); the DoG
        // subtraction is fused into the blur launches, so its own time is zero
        const double blur = t.ms[SIFT_STAGE_BLUR_L1] + t.ms[SIFT_STAGE_BLUR_L2] + t.ms[SIFT_STAGE_BLUR_L3] +
                            t.ms[SIFT_STAGE_BLUR_L4] + t.ms[SIFT_STAGE_BLUR_L5];
        const double extrema = t.ms[SIFT_STAGE_GRADIENT] + t.ms[SIFT_STAGE_EXTREMA] + t.ms[SIFT_STAGE_LOCALIZE];
        std::cerr << "elapsed time: gen_gaussian_images: " << 1e-3 * blur << "s\n";
        std::cerr << "elapsed time: gen_dog_images: " << 0.0 << "s\n";
        std::cerr << "elapsed time: gen_scale_space_extrema: " << 1e-3 * extrema << "s\n";
        std::cerr << "elapsed time: clean_keypoints: " << 1e-3 * t.ms[SIFT_STAGE_CLEAN] << "s\n";
        std::cerr << "elapsed time: get_descriptors: " << 1e-3 * t.ms[SIFT_STAGE_DESCRIPTOR] << "s\n";

        const int n = counts.keypoints;
        std::vector<sift_keypoint> kps((size_t)n);
        std::vector<double> desc((size_t)n * 128);
        int32_t got = 0;
        check(sift_batch_download(batch, 0, n, kps.data(), nullptr, desc.data(), &got), "sift_batch_download");
        keypoints.clear();
        descriptors.clear();
        for (int i = 0; i < n; i++) {
            keypoints.emplace_back(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave,
                                   kps[i].class_id);
            descriptors.emplace_back(desc.begin() + (size_t)i * 128, desc.begin() + (size_t)(i + 1) * 128);
        }
        base = cv::Mat(2 * height_, 2 * width_, CV_64FC1);  // sift.cpp:45
        check(sift_batch_get_level(batch, 0, 0, 0, SIFT_LEVEL_GAUSS, base.ptr<double>(0)), "sift_batch_get_level");
    }
    std::cerr << "Keypoints Calculated: " << keypoints.size() << "\n";
    dump_keypoints();
    std::cerr << "Completed SIFT for " << name << "\n\n";
}

// sift.cpp:103-112
cv::Mat sift_handler::get() const {
    cv::Mat desc((int)descriptors.size(), 128, CV_32FC1);
    for (int i = 0; i < (int)descriptors.size(); ++i)
        for (int j = 0; j < (int)descriptors[i].size(); ++j) desc.at<float>(i, j) = (float)descriptors[i][j];
    return desc;
}

// sift.cpp:114-121
void sift_handler::dump_keypoints() {
    std::ofstream fout;
    fout.open(this->name + ".csv", std::ios::out);
    // same bytes as the reference's per-line std::endl, written with one flush at the end
    for (auto &kpt : this->keypoints) fout << kpt.pt.x << "," << kpt.pt.y << "," << kpt.size << "," << kpt.angle << "\n";
    fout.close();
}

std::vector<match_result> knn_match(const cv::Mat &query, const cv::Mat &train) {
    if ((query.rows && (query.cols != 128 || query.type() != CV_32FC1)) ||
        (train.rows && (train.cols != 128 || train.type() != CV_32FC1)))
        throw std::runtime_error("knn_match: expected N x 128 CV_32FC1 descriptor matrices");
    const int nq = query.rows, nt = train.rows;
    std::vector<float> q((size_t)nq * 128), t((size_t)nt * 128);
    for (int i = 0; i < nq; i++) std::memcpy(q.data() + (size_t)i * 128, query.ptr<float>(i), 128 * sizeof(float));
    for (int i = 0; i < nt; i++) std::memcpy(t.data() + (size_t)i * 128, train.ptr<float>(i), 128 * sizeof(float));
    std::vector<sift_knn2> knn((size_t)nq);
    std::vector<uint8_t> accept((size_t)nq);
    check(sift_knn2_f32(context(), q.data(), nq, t.data(), nt, knn.data(), accept.data()), "sift_knn2_f32");
    std::vector<match_result> out((size_t)nq);
    for (int i = 0; i < nq; i++) {
        match_result &m = out[i];
        m.queryIdx = i;
        m.trainIdx = knn[i].i1;
        m.trainIdx2 = knn[i].i2;
        m.distance = knn[i].i1 >= 0 ? std::sqrt((float)knn[i].d1) : 0.f;
        m.distance2 = knn[i].i2 >= 0 ? std::sqrt((float)knn[i].d2) : 0.f;
        m.ratio_ok = accept[i] != 0;
    }
    return out;
}

std::vector<match_result> good_matches(const std::vector<match_result> &matches) {
    double min_dist = 100;  // main.cpp:77
    for (const auto &m : matches)
        if (m.trainIdx >= 0 && m.distance < min_dist) min_dist = m.distance;
    std::vector<match_result> good;
    for (const auto &m : matches)
        if (m.trainIdx >= 0 && m.distance < 3 * min_dist) good.push_back(m);  // main.cpp:88
    return good;
}

homography_result find_homography(const std::vector<cv::Point2f> &obj, const std::vector<cv::Point2f> &scene,
                                  double ransacReprojThreshold, int maxIters, double confidence) {
    if (obj.size() != scene.size()) throw std::runtime_error("find_homography: obj and scene differ in length");
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "Point2f is an (x, y) float pair");
    const int n = (int)obj.size();
    homography_result r;
    r.mask.assign((size_t)n, 0);
    sift_homography_info info{};
    check(sift_find_homography(context(), n ? &obj[0].x : nullptr, n ? &scene[0].x : nullptr, n, ransacReprojThreshold, maxIters,
                               confidence, r.H, r.mask.data(), &info),
          "sift_find_homography");
    r.found = info.found != 0;
    r.iterations = info.iterations;
    r.inliers = info.inliers;
    return r;
}

static void require_bgr8(const cv::Mat &m, const char *what) {
    if (m.empty() || m.type() != CV_8UC3) throw std::runtime_error(std::string(what) + ": expected a non-empty CV_8UC3 image");
}

cv::Mat stitch_pair(const cv::Mat &ref, const cv::Mat &img, const double H[9]) {
    require_bgr8(ref, "stitch_pair");
    require_bgr8(img, "stitch_pair");
    std::vector<unsigned char> buf((size_t)ref.rows * ((size_t)ref.cols + img.cols) * 3);
    int w = 0;
    check(sift_stitch_pair_bgr8(context(), ref.data, ref.cols, ref.rows, ref.step, img.data, img.cols, img.rows, img.step, H,
                                buf.data(), &w),
          "sift_stitch_pair_bgr8");
    cv::Mat out(ref.rows, w, CV_8UC3);
    for (int r = 0; r < ref.rows && w > 0; r++) std::memcpy(out.ptr<unsigned char>(r), buf.data() + (size_t)r * w * 3, (size_t)w * 3);
    return out;
}

cv::Mat warp_perspective(const cv::Mat &src, const double H[9], cv::Size dsize) {
    require_bgr8(src, "warp_perspective");
    cv::Mat out(dsize.height, dsize.width, CV_8UC3);
    if (out.step != (size_t)dsize.width * 3) throw std::runtime_error("warp_perspective: unexpected row padding");
    check(sift_warp_perspective_bgr8(context(), src.data, src.cols, src.rows, src.step, H, dsize.width, dsize.height, out.data),
          "sift_warp_perspective_bgr8");
    return out;
}

}  // namespace sift
