// Wall time of sift_extract_gray8 from C++ (no Python in the loop), for a synthetic frame.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sift_amd.h"

int main(int argc, char **argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 512, h = argc > 2 ? atoi(argv[2]) : 512;
    std::vector<uint8_t> img((size_t)w * h);
    sift_synth_dead_leaves(3, 0, w, h, img.data());
    sift_ctx *ctx;
    if (sift_ctx_create(0, &ctx)) return 1;
    sift_result r;
    for (int k = 0; k < 3; k++) {
        if (sift_extract_gray8(ctx, img.data(), w, h, w, &r)) return 2;
        sift_result_free(&r);
    }
    const int reps = 50;
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    for (int k = 0; k < reps; k++) {
        sift_extract_gray8(ctx, img.data(), w, h, w, &r);
        n = r.n;
        sift_result_free(&r);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%dx%d keypoints %d sift_extract_gray8 %.3f ms\n", w, h, n, ms);
    sift_ctx_destroy(ctx);
    return 0;
}
