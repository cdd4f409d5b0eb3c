// Synthetic inputs for benchmarks and tests (host code, integer-only so the
// same bytes come out on every machine).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "sift_amd.h"

namespace {

constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;

struct SplitMix64 {
    uint64_t state;
    uint64_t next() {
        state += GOLDEN;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};

void descriptor_row(uint64_t seed, uint64_t row, uint8_t *out) {
    SplitMix64 g{seed * GOLDEN + row};
    for (int k = 0; k < 128; k += 4) {
        uint64_t r = g.next();
        for (int e = 0; e < 4; e++) {
            const uint32_t a = (uint32_t)(r & 0xff), b = (uint32_t)((r >> 8) & 0xff);
            r >>= 16;
            const uint32_t v = (a * b * b) >> 15;
            out[k + e] = (uint8_t)(v > 102 ? 102 : v);
        }
    }
}

template <class F>
void parallel_rows(int64_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / 4096));
    if (nt <= 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(f, n * t / nt, n * (t + 1) / nt);
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" {

int sift_synth_dead_leaves(uint64_t seed, uint64_t frame_index, int w, int h, uint8_t *gray) {
    if (!gray || w <= 0 || h <= 0) return SIFT_ERR_INVALID;
    std::memset(gray, 128, (size_t)w * h);
    SplitMix64 g{seed * GOLDEN + frame_index};
    const int64_t discs = (int64_t)w * h / 128;
    for (int64_t k = 0; k < discs; k++) {
        const int cx = (int)(g.next() % (uint64_t)w);
        const int cy = (int)(g.next() % (uint64_t)h);
        const int e = (int)(g.next() % 64);
        const uint8_t v = (uint8_t)(g.next() % 256);
        const int r = 2 + (e * e * e) / 4096;
        const int y0 = std::max(0, cy - r), y1 = std::min(h - 1, cy + r);
        for (int y = y0; y <= y1; y++) {
            const int dy = y - cy;
            const int x0 = std::max(0, cx - r), x1 = std::min(w - 1, cx + r);
            uint8_t *row = gray + (size_t)y * w;
            for (int x = x0; x <= x1; x++) {
                const int dx = x - cx;
                if (dx * dx + dy * dy <= r * r) row[x] = v;
            }
        }
    }
    return SIFT_OK;
}

int sift_synth_descriptors(uint64_t seed, int64_t n, uint8_t *out) {
    if (!out || n < 0) return SIFT_ERR_INVALID;
    parallel_rows(n, [=](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; i++) descriptor_row(seed, (uint64_t)i, out + (size_t)i * 128);
    });
    return SIFT_OK;
}

int sift_synth_queries(uint64_t seed, const uint8_t *train, int64_t nt, int64_t nq, uint8_t *out) {
    if (!out || nq < 0 || nt < 0 || (nt > 0 && !train)) return SIFT_ERR_INVALID;
    parallel_rows(nq, [=](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; i++) {
            uint8_t *dst = out + (size_t)i * 128;
            if ((i & 1) == 0 && nt > 0) {
                SplitMix64 g{seed * GOLDEN + (uint64_t)i};
                const uint8_t *src = train + (size_t)(g.next() % (uint64_t)nt) * 128;
                for (int k = 0; k < 128; k += 16) {
                    uint64_t r = g.next();
                    for (int q = 0; q < 16; q++) {
                        const int noise = (int)(r % 5) - 2;
                        r >>= 4;
                        int v = (int)src[k + q] + noise;
                        v = v < 0 ? 0 : (v > 102 ? 102 : v);
                        dst[k + q] = (uint8_t)v;
                    }
                }
            } else {
                descriptor_row(seed ^ 0x5151515151515151ull, (uint64_t)i, dst);
            }
        }
    });
    return SIFT_OK;
}

}  // extern "C"
