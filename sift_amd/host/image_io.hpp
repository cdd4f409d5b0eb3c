// Image files -> cv::Mat the way cv::imread(path) (IMREAD_COLOR) hands them to
// the reference's main.cpp:30, :55, :67: 8-bit, 3 channels, BGR order, alpha
// dropped, grey replicated, 16-bit samples reduced to their high byte.
// Supported: PNG (non-interlaced; decoded here on top of zlib), JPEG (Huffman
// baseline and progressive, decoded here with libjpeg-turbo's integer
// arithmetic; EXIF orientation applied) and binary PGM/PPM.  An unreadable file
// yields an empty Mat, like cv::imread.
#pragma once
#include <string>

#include "sift.hpp"

namespace sift_io {

cv::Mat imread(const std::string &path);

// Decodes a PNG held in memory; `error`, when given, receives the reason for an empty result.
cv::Mat decode_png(const unsigned char *data, std::size_t size, std::string *error = nullptr);

// cv::imwrite(path, bgr) for ".png" paths as main.cpp:112, :149 uses it: an 8-bit RGB PNG of the
// CV_8UC3 (BGR) image.  The pixels read back are the ones written; the byte stream is this encoder's own
// (one IDAT, filter type 0, zlib default level).  Returns false when the file cannot be written.
bool imwrite(const std::string &path, const cv::Mat &bgr);

// Decodes a JPEG held in memory (jpeg_decode.cpp).
cv::Mat decode_jpeg(const unsigned char *data, std::size_t size, std::string *error = nullptr);

}  // namespace sift_io
