// Prints width, height, the BGR byte sum and an FNV-1a hash of the decoded pixels of each file given.
// With "--copy <in> <out.png>": reads <in> and writes it back with sift_io::imwrite.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "image_io.hpp"

int main(int argc, char **argv) {
    if (argc == 4 && !std::strcmp(argv[1], "--copy")) {
        cv::Mat m = sift_io::imread(argv[2]);
        return m.data && sift_io::imwrite(argv[3], m) ? 0 : 1;
    }
    for (int k = 1; k < argc; k++) {
        cv::Mat m = sift_io::imread(argv[k]);
        if (!m.data) {
            printf("%s EMPTY\n", argv[k]);
            continue;
        }
        uint64_t sum = 0, hash = 1469598103934665603ull;
        const size_t n = (size_t)m.rows * m.cols * 3;
        for (size_t i = 0; i < n; i++) {
            sum += m.data[i];
            hash = (hash ^ m.data[i]) * 1099511628211ull;
        }
        printf("%s %d %d %llu %llu\n", argv[k], m.cols, m.rows, (unsigned long long)sum, (unsigned long long)hash);
    }
    return 0;
}
