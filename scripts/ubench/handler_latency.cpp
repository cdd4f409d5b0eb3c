// Wall time of sift::sift_handler construction + exec() + get() (the reference's class surface) for a synthetic frame.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sift.hpp"
#include "sift_amd.h"

int main(int argc, char **argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 512, h = argc > 2 ? atoi(argv[2]) : 512;
    cv::Mat img(h, w, CV_8UC1);
    sift_synth_dead_leaves(3, 0, w, h, img.data);
    FILE *devnull = freopen("/dev/null", "w", stderr);
    (void)devnull;
    size_t n = 0;
    double ms[3] = {0, 0, 0};
    const int reps = 12;
    for (int k = 0; k < reps + 2; k++) {
        auto t0 = std::chrono::steady_clock::now();
        sift::sift_handler s("latency", img.clone());
        auto t1 = std::chrono::steady_clock::now();
        s.exec();
        auto t2 = std::chrono::steady_clock::now();
        cv::Mat d = s.get();
        auto t3 = std::chrono::steady_clock::now();
        n = s.keypoints.size();
        if (k >= 2) {
            ms[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            ms[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
            ms[2] += std::chrono::duration<double, std::milli>(t3 - t2).count();
        }
    }
    printf("%dx%d keypoints %zu  constructor %.2f ms  exec() %.2f ms  get() %.2f ms\n", w, h, n, ms[0] / reps, ms[1] / reps, ms[2] / reps);
    return 0;
}
