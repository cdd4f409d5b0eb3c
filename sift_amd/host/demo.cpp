// Counterpart of the reference's src/main.cpp up to the match list
// (main.cpp:22-93): extract features of every image, then match the last image
// (the reference) against each of the others and report the good matches.
// Images are PNG or binary PGM/PPM files, read by sift_io::imread with the
// conventions of cv::imread (BGR, 8-bit, 3 channels).
//
//   sift_demo <image>                 -> keypoints only (main.cpp:28-39)
//   sift_demo <img1> ... <imgN>       -> match each img against imgN
#include <cstdio>
#include <iostream>
#include <sstream>

#include "image_io.hpp"
#include "sift.hpp"

// "dir/name.ext" -> "name" (what main.cpp:13-19 passes to the handler as its name)
static std::string stem_of(const std::string &path) {
    const std::size_t slash = path.rfind('/');
    const std::string file = slash == std::string::npos ? path : path.substr(slash + 1);
    return file.substr(0, file.rfind('.'));
}

int main(int argc, char **argv) {
    if (argc < 2) {
        printf("usage: sift_demo <Image_Path> <Image_Path>\n");
        return -1;
    }
    if (argc == 2) {
        cv::Mat image = sift_io::imread(argv[1]);
        if (!image.data) {
            printf("No image data \n");
            return -1;
        }
        sift::sift_handler ss(stem_of(argv[1]), std::move(image));
        ss.exec();
        return 0;
    }
    cv::Mat ref = sift_io::imread(argv[argc - 1]);
    if (!ref.data) {
        printf("No image data \n");
        return -1;
    }
    const std::string name_ref = stem_of(argv[argc - 1]);
    sift::sift_handler s_ref(name_ref, ref.clone());
    s_ref.exec();
    cv::Mat ref_descriptors = s_ref.get();
    for (int k = argc - 2; k >= 1; k--) {
        cv::Mat img = sift_io::imread(argv[k]);
        if (!img.data) {
            printf("No image data - %s \n", argv[k]);
            return -1;
        }
        sift::sift_handler s(stem_of(argv[k]), std::move(img));
        s.exec();
        cv::Mat desc = s.get();
        auto matches = sift::knn_match(ref_descriptors, desc);  // query = ref, train = img (main.cpp:75)
        auto good = sift::good_matches(matches);
        int ratio = 0;
        for (const auto &m : matches) ratio += m.ratio_ok;
        std::cerr << "Good Matches found:" << good.size() << std::endl;  // main.cpp:93
        std::cout << "matches " << matches.size() << " good " << good.size() << " ratio_test " << ratio << std::endl;
    }
    return 0;
}
