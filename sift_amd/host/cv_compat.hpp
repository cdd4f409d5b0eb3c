// Minimal stand-ins for the four OpenCV types that sift.hpp exposes
// (cv::Mat, cv::KeyPoint, cv::Point2f, cv::Size), used only when the real
// OpenCV headers are not installed.  With OpenCV present, sift.hpp includes
// <opencv2/core.hpp> instead and this file is not used.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <ostream>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

template <typename T>
struct Point_ {
    T x{}, y{};
    Point_() = default;
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
using Point2f = Point_<float>;

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};
inline std::ostream &operator<<(std::ostream &os, const Size &s) { return os << "[" << s.width << " x " << s.height << "]"; }

// Same members, order and widths as cv::KeyPoint.
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() = default;
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

// Large matrix blocks are recycled: a fresh 66 MB allocation (the base image of a 1080p frame) is mapped and
// page-faulted by the kernel every time, which costs several times the GPU work of the frame; a released block is
// kept (up to 512 MB in all) and handed to the next matrix of at most its size.
namespace detail {
struct BlockCache {
    struct Block {
        unsigned char *p;
        size_t bytes;
    };
    std::mutex mu;
    std::vector<Block> blocks;
    size_t held = 0;
    static constexpr size_t kMinBytes = (size_t)1 << 20, kMaxHeld = (size_t)512 << 20;
    // Never destroyed: a matrix with static storage may give its block back after every other static object of
    // the program is gone.
    static BlockCache &get() {
        static BlockCache *c = new BlockCache;
        return *c;
    }
    // frees the held blocks (a long-lived process that has finished with large images)
    void trim() {
        std::lock_guard<std::mutex> lock(mu);
        for (const Block &b : blocks) delete[] b.p;
        blocks.clear();
        held = 0;
    }
    unsigned char *take(size_t bytes, size_t *got) {
        if (bytes >= kMinBytes) {
            std::lock_guard<std::mutex> lock(mu);
            size_t best = blocks.size();
            for (size_t i = 0; i < blocks.size(); i++)
                if (blocks[i].bytes >= bytes && blocks[i].bytes <= 2 * bytes && (best == blocks.size() || blocks[i].bytes < blocks[best].bytes))
                    best = i;
            if (best != blocks.size()) {
                const Block b = blocks[best];
                blocks.erase(blocks.begin() + (std::ptrdiff_t)best);
                held -= b.bytes;
                *got = b.bytes;
                return b.p;
            }
        }
        *got = bytes;
        return new unsigned char[bytes];
    }
    void give(unsigned char *p, size_t bytes) {
        if (bytes >= kMinBytes) {
            std::lock_guard<std::mutex> lock(mu);
            if (held + bytes <= kMaxHeld) {
                blocks.push_back(Block{p, bytes});
                held += bytes;
                return;
            }
        }
        delete[] p;
    }
};
}  // namespace detail

// Releases the matrix blocks kept for reuse (no counterpart in OpenCV).
inline void trimMatCache() { detail::BlockCache::get().trim(); }

// Dense 2-D matrix with reference-counted storage (the subset of cv::Mat used
// around sift_handler: construction, at<>, ptr<>, clone, release, size, type).
class Mat {
   public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    size_t step = 0;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    // wraps caller memory without copying (the caller keeps ownership)
    Mat(int r, int c, int type, void *external, size_t step_bytes = 0)
        : rows(r), cols(c), data(static_cast<unsigned char *>(external)), type_(type) {
        step = step_bytes ? step_bytes : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elemSize();
        // contents are indeterminate, as with cv::Mat::create
        size_t got = 0;
        unsigned char *block = detail::BlockCache::get().take(step * (size_t)(r > 0 ? r : 1), &got);
        store_ = std::shared_ptr<unsigned char>(block, [got](unsigned char *q) { detail::BlockCache::get().give(q, got); });
        data = store_.get();
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const {
        static const int bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        return (size_t)bytes[depth()] * channels();
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    void release() {
        store_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    Mat clone() const {
        Mat m;
        if (!data) return m;
        m.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + r * m.step, data + r * step, (size_t)cols * elemSize());
        return m;
    }
    template <typename T>
    T &at(int r, int c) { return *reinterpret_cast<T *>(data + r * step + c * sizeof(T)); }
    template <typename T>
    const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + r * step + c * sizeof(T)); }
    template <typename T>
    T *ptr(int r = 0) { return reinterpret_cast<T *>(data + r * step); }
    template <typename T>
    const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + r * step); }

   private:
    int type_ = 0;
    std::shared_ptr<unsigned char> store_;
};

}  // namespace cv
