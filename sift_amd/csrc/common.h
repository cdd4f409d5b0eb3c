// Shared host/device definitions of the MI355X SIFT engine.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sift_amd.h"

namespace sa {

// Tunables of the reference, sift.hpp:97-112 (compile-time constants there too).
constexpr int SCALES = 3;
constexpr int BORDER = 5;
constexpr double CONTRAST_THRESHOLD = 0.04;
constexpr double SIGMA = 1.6;
constexpr double ASSUMED_BLUR = 0.5;
constexpr int IMAGES = SCALES + 3;
constexpr int NDOG = IMAGES - 1;
constexpr double EIGEN_VALUE_RATIO = 10.0;
constexpr int BINS = 36;
constexpr double PEAK_RATIO = 0.8;
constexpr double SCALE_FACTOR = 1.5;
constexpr double RADIUS_FACTOR = 3.0;
constexpr double PI = 3.14159265358979323846;
constexpr double SCALE_MULTIPLIER = 3.0;
constexpr int WINDOW_WIDTH = 4;
constexpr double DESCRIPTOR_MAX = 0.2;
constexpr double EPS = 1e-7;   // sift.cpp:17
constexpr double EPS2 = .0001; // sift.cpp:18

constexpr int MAX_OCTAVES = 16;
constexpr int MAX_TAPS = 27;

// Symmetric Gaussian taps, passed to kernels by value (lands in SGPRs).
struct Taps {
    double c[MAX_TAPS];
};

// Gradient of a Gaussian level at one pixel: magnitude and direction in degrees.
struct alignas(16) Grad {
    double mag;
    double ori;
};

// One frame batch's scale space.  gauss[o] is [frame][IMAGES][h][w] doubles,
// rows tightly packed (pitch = w).  The difference-of-Gaussians level j is
// gauss level j+1 minus gauss level j (sift.cpp:173), formed in registers by its readers.
// grad[o] is [frame][SCALES][h][w] Grad: the gradient planes of Gaussian
// levels 1..SCALES, the only levels orientation and descriptor windows sample
// (null for octaves too small to hold a keypoint).
struct Pyramid {
    int octaves;
    int w[MAX_OCTAVES];
    int h[MAX_OCTAVES];
    double *gauss[MAX_OCTAVES];
    Grad *grad[MAX_OCTAVES];
};

__host__ __device__ inline size_t gauss_off(const Pyramid &p, int o, int frame, int lvl) {
    return ((size_t)frame * IMAGES + lvl) * (size_t)p.w[o] * p.h[o];
}
// layer = Gaussian level 1..SCALES
__host__ __device__ inline size_t grad_off(const Pyramid &p, int o, int frame, int layer) {
    return ((size_t)frame * SCALES + (layer - 1)) * (size_t)p.w[o] * p.h[o];
}

// Device keypoint record: a sift_keypoint padded to 32 bytes.
struct DevKeypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id, pad;
};

// Candidate pixel of the extrema scan.
struct Candidate {
    int32_t oct, img, i, j;
};

// Per-frame counters living in device memory.
struct FrameCounters {
    int32_t candidates;
    int32_t oriented;
    int32_t keypoints;
    int32_t overflow;
};

// borderInterpolate(p, len, BORDER_REFLECT_101), repeated until inside.
__host__ __device__ inline int reflect101(int p, int len) {
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

#ifdef __HIPCC__
// Workgroups of a launch are handed to the 8 XCDs round-robin in launch order
// (x fastest), and every XCD has its own L2.  Image kernels whose neighbouring
// tiles share halo lines therefore renumber their workgroups so that each XCD
// works through one contiguous run of tiles: neighbours are then resident on
// the same XCD at about the same time and the shared lines are L2 hits instead
// of second HBM fetches.  This is a bijection on the grid whatever the real
// placement is; only the hit rate depends on it.
struct TileId {
    int x, y, z;
};
__device__ __forceinline__ TileId xcd_tile_id() {
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const unsigned total = gx * gy * gz;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, k = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned n = xcd * q + (xcd < r ? xcd : r) + k;
    TileId t;
    t.x = (int)(n % gx);
    t.y = (int)((n / gx) % gy);
    t.z = (int)(n / (gx * gy));
    return t;
}
#endif

}  // namespace sa
