// Counterpart of the reference's src/main.cpp (main.cpp:22-152): extract features of every image,
// then, from the last image backwards,
// match the running panorama against the next image, report the good matches,
// estimate the homography that maps the panorama onto the image, compose the
// two ("Final-<names>.png") and carry the result on as the new panorama.  Like the reference it
// writes "<name>_keypoints.png" per extraction (sift.cpp:89-90) and "Matches<names>.png" per pair
// (main.cpp:110-112); SIFT_AMD_KEYPOINT_PNG=0 turns the former off.
// Images are PNG or binary PGM/PPM files, read by sift_io::imread with the
// conventions of cv::imread (BGR, 8-bit, 3 channels).
//
//   sift_demo <image>                 -> keypoints only (main.cpp:28-39)
//   sift_demo <img1> ... <imgN>       -> match each img against imgN
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <vector>

#include "image_io.hpp"
#include "sift.hpp"

// "dir/name.ext" -> "name" (what main.cpp:13-19 passes to the handler as its name)
static std::string stem_of(const std::string &path) {
    const std::size_t slash = path.rfind('/');
    const std::string file = slash == std::string::npos ? path : path.substr(slash + 1);
    return file.substr(0, file.rfind('.'));
}

int main(int argc, char **argv) {
    {
        const char *env = std::getenv("SIFT_AMD_KEYPOINT_PNG");
        sift::write_keypoint_images(!(env && env[0] == '0'));
    }
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
    std::string name_ref = stem_of(argv[argc - 1]);
    for (int k = argc - 2; k >= 1; k--) {
        sift::sift_handler s_ref(name_ref, ref.clone());  // main.cpp:56-60: the panorama so far is extracted again
        s_ref.exec();
        cv::Mat ref_descriptors = s_ref.get();
        cv::Mat img = sift_io::imread(argv[k]);
        if (!img.data) {
            printf("No image data - %s \n", argv[k]);
            return -1;
        }
        const std::string name = stem_of(argv[k]);
        sift::sift_handler s(name, img.clone());
        s.exec();
        cv::Mat desc = s.get();
        auto matches = sift::knn_match(ref_descriptors, desc);  // query = ref, train = img (main.cpp:75)
        auto good = sift::good_matches(matches);
        int ratio = 0;
        for (const auto &m : matches) ratio += m.ratio_ok;
        std::cerr << "Good Matches found:" << good.size() << std::endl;  // main.cpp:93
        std::cout << "matches " << matches.size() << " good " << good.size() << " ratio_test " << ratio << std::endl;
        std::vector<cv::Point2f> obj, scene;  // main.cpp:92-101
        for (const auto &gm : good) {
            obj.push_back(s_ref.keypoints[gm.queryIdx].pt);
            scene.push_back(s.keypoints[gm.trainIdx].pt);
        }
        const auto hr = sift::find_homography(obj, scene, 3);  // main.cpp:105
        std::cout << "homography found " << hr.found << " inliers " << hr.inliers << " iterations " << hr.iterations << std::endl;
        if (hr.found) {
            std::cout.precision(17);
            for (int r = 0; r < 3; r++) std::cout << "H " << hr.H[3 * r] << " " << hr.H[3 * r + 1] << " " << hr.H[3 * r + 2] << std::endl;
            std::cout.precision(6);
        }
        name_ref += '+' + name;  // main.cpp:108
        {   // main.cpp:109-112: drawMatches(ref, ref_kpt, img, kpt, good_matches, ...) + imwrite
            std::vector<std::pair<int, int>> pairs;
            for (const auto &gm : good) pairs.emplace_back(gm.queryIdx, gm.trainIdx);
            sift_io::imwrite("./Matches" + name_ref + ".png", sift_io::draw_matches(ref, s_ref.keypoints, img, s.keypoints, pairs));
        }
        if (!hr.found || img.rows > ref.rows) {
            std::cout << "no panorama for " << name_ref << std::endl;  // the reference throws here
            return 1;
        }
        cv::Mat result = sift::stitch_pair(ref, img, hr.H);  // main.cpp:115-145
        std::cerr << result.size() << "\n";                   // main.cpp:147
        if (!sift_io::imwrite("./Final-" + name_ref + ".png", result)) return 1;  // main.cpp:149
        std::cout << "panorama " << result.cols << " " << result.rows << std::endl;
        ref = std::move(result);  // main.cpp:151
    }
    return 0;
}
