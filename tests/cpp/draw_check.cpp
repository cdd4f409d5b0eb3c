// Host check of the overlays that stand in for cv::drawKeypoints / cv::drawMatches (sift.cpp:89, main.cpp:110):
// draws keypoints that lie inside, across and far outside the canvas and writes both pictures to argv[1], argv[2].
#include "image_io.hpp"
#include <cstdio>
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    cv::Mat g(120, 160, CV_8UC1);
    for (int y = 0; y < 120; y++) for (int x = 0; x < 160; x++) g.ptr<unsigned char>(y)[x] = (unsigned char)(x + y);
    std::vector<cv::KeyPoint> k = {cv::KeyPoint(40, 50, 30, 45), cv::KeyPoint(150, 110, 60, 200), cv::KeyPoint(5, 5, 400, 10), cv::KeyPoint(80, 60, 0, -1)};
    cv::Mat o = sift_io::draw_keypoints(g, k);
    if (!sift_io::imwrite(argv[1], o)) return 1;
    cv::Mat a = o.clone();
    cv::Mat m = sift_io::draw_matches(a, k, o, k, {{0, 1}, {1, 0}, {2, 3}, {9, 9}});
    if (!sift_io::imwrite(argv[2], m)) return 1;
    printf("%d %d %d %d\n", o.cols, o.rows, m.cols, m.rows);
}
