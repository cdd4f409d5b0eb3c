// sift::sift_handler over the C ABI (see sift.hpp).  Mirrors the control flow
// and side effects of the reference's src/sift.cpp constructor, exec(), get()
// and dump_keypoints(); every numeric stage runs in libsift_amd.so.
#include "sift.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <stdexcept>

#include "image_io.hpp"
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

// -1: not decided yet (the environment is consulted on first use)
int g_keypoint_png = -1;

bool keypoint_png_enabled() {
    if (g_keypoint_png < 0) {
        const char *env = std::getenv("SIFT_AMD_KEYPOINT_PNG");
        g_keypoint_png = env && env[0] == '1';
    }
    return g_keypoint_png != 0;
}

}  // namespace

bool write_keypoint_images(bool on) {
    const bool before = keypoint_png_enabled();
    g_keypoint_png = on;
    return before;
}

// sift.cpp:26-46: gray conversion, onex, octave count, and the base image (2x upsample + blur, on the GPU).
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
    for (int y = 0; y < height_; y++) {
        double *dst = onex.ptr<double>(y);
        const unsigned char *src = gray_.data() + (size_t)y * width_;
        for (int x = 0; x < width_; x++) dst[x] = src[x];
    }
    octaves = sift_num_octaves(width_, height_);  // sift.cpp:36
    std::cerr << "Octaves: " << octaves << std::endl;
    std::cerr << "Size: " << cv::Size(width_, height_) << std::endl;
    base = cv::Mat(2 * height_, 2 * width_, CV_64FC1);  // sift.cpp:40-45
    if (base.step != (size_t)2 * width_ * sizeof(double)) throw std::runtime_error("sift_handler: unexpected row padding");
    check(sift_base_image_gray8(context(), gray_.data(), width_, height_, (size_t)width_, base.ptr<double>(0)),
          "sift_base_image_gray8");
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
    // the whole of exec() is one library call on the fast single-image schedule; keypoints and descriptors are
    // unbounded vectors in the reference (sift.hpp:93-94), and the call grows its workspace when a frame needs it
    sift_result res;
    double ms[5];
    check(sift_extract_gray8_stages(context(), gray_.data(), width_, height_, (size_t)width_, &res, ms),
          "sift_extract_gray8_stages");
    // the reference's five TIMEIT stages (sift.cpp:71,72,86,87,92)
    static const char *const stage[5] = {"gen_gaussian_images", "gen_dog_images", "gen_scale_space_extrema", "clean_keypoints",
                                         "get_descriptors"};
    for (int i = 0; i < 5; i++) std::cerr << "elapsed time: " << stage[i] << ": " << 1e-3 * ms[i] << "s\n";
    keypoints.clear();
    descriptors.clear();
    keypoints.reserve((size_t)res.n);
    descriptors.reserve((size_t)res.n);
    for (int i = 0; i < res.n; i++) {
        const sift_keypoint &k = res.keypoints[i];
        keypoints.emplace_back(k.x, k.y, k.size, k.angle, k.response, k.octave, k.class_id);
        descriptors.emplace_back(res.descriptors + (size_t)i * 128, res.descriptors + (size_t)(i + 1) * 128);
    }
    sift_result_free(&res);
    if (keypoint_png_enabled()) {  // sift.cpp:89-90: drawKeypoints(onex as 8-bit, keypoints, DRAW_RICH_KEYPOINTS) + imwrite
        cv::Mat gray(height_, width_, CV_8UC1, gray_.data());
        sift_io::imwrite(name + "_keypoints.png", sift_io::draw_keypoints(gray, keypoints));
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
    // Same bytes as the reference's `fout << x << "," << y << "," << size << "," << angle << std::endl`: a float goes
    // through the stream as "%g" with six significant digits, which is what to_chars(general, 6) produces; the lines
    // are assembled in memory and written once.
    std::string text;
    text.reserve(this->keypoints.size() * 48);
    char buf[32];
    auto put = [&](float v, char sep) {
        const auto r = std::to_chars(buf, buf + sizeof(buf), (double)v, std::chars_format::general, 6);
        text.append(buf, r.ptr);
        text.push_back(sep);
    };
    for (auto &kpt : this->keypoints) {
        put(kpt.pt.x, ',');
        put(kpt.pt.y, ',');
        put(kpt.size, ',');
        put(kpt.angle, '\n');
    }
    std::ofstream fout;
    fout.open(this->name + ".csv", std::ios::out | std::ios::binary);
    fout.write(text.data(), (std::streamsize)text.size());
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
