// Image files -> cv::Mat the way cv::imread(path) (IMREAD_COLOR) hands them to
// the reference's main.cpp:30, :55, :67: 8-bit, 3 channels, BGR order, alpha
// dropped, grey replicated, 16-bit samples reduced to their high byte.
// Supported: PNG (non-interlaced; decoded here on top of zlib), JPEG (Huffman
// baseline and progressive, decoded here with libjpeg-turbo's integer
// arithmetic; EXIF orientation applied) and binary PGM/PPM.  An unreadable file
// yields an empty Mat, like cv::imread.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "sift.hpp"

namespace sift_io {

cv::Mat imread(const std::string &path);

// Decodes a PNG held in memory; `error`, when given, receives the reason for an empty result.
cv::Mat decode_png(const unsigned char *data, std::size_t size, std::string *error = nullptr);

// cv::imwrite(path, bgr) for ".png" paths as main.cpp:112, :149 uses it: an 8-bit RGB PNG of the
// CV_8UC3 (BGR) image.  The pixels read back are the ones written; the byte stream is this encoder's own
// (one IDAT, filter type 0, zlib default level).  Returns false when the file cannot be written.
bool imwrite(const std::string &path, const cv::Mat &bgr);

// Overlay in the manner of cv::drawKeypoints(gray, keypoints, out, Scalar::all(-1), DRAW_RICH_KEYPOINTS)
// (sift.cpp:89): the 8-bit gray (or BGR) image as BGR with, per keypoint, a circle of radius size/2 and a radius
// along its angle, one colour per keypoint.  OpenCV draws anti-aliased with colours from its global generator;
// this rasteriser (midpoint circle, Bresenham line) and its colour sequence are its own, so the files look alike
// and are not comparable byte for byte.
cv::Mat draw_keypoints(const cv::Mat &image, const std::vector<cv::KeyPoint> &keypoints);

// Overlay in the manner of cv::drawMatches(a, ka, b, kb, matches, out, ..., NOT_DRAW_SINGLE_POINTS) (main.cpp:110):
// the two BGR images side by side, and for every (index into ka, index into kb) pair a line between the two
// keypoints with a small circle at either end.
cv::Mat draw_matches(const cv::Mat &a, const std::vector<cv::KeyPoint> &ka, const cv::Mat &b,
                     const std::vector<cv::KeyPoint> &kb, const std::vector<std::pair<int, int>> &pairs);

// Decodes a JPEG held in memory (jpeg_decode.cpp).
cv::Mat decode_jpeg(const unsigned char *data, std::size_t size, std::string *error = nullptr);

}  // namespace sift_io
