#include "image_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

namespace sift_io {

namespace {

uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

cv::Mat fail(std::string *error, const char *why) {
    if (error) *error = why;
    return cv::Mat();
}

// PNG filter type 4 predictor
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

cv::Mat read_pnm(const std::vector<unsigned char> &buf) {
    // header: magic, width, height, maxval separated by whitespace, '#' comments allowed
    std::size_t pos = 2;
    auto number = [&]() -> long {
        for (;;) {
            while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\t' || buf[pos] == '\n' || buf[pos] == '\r')) pos++;
            if (pos < buf.size() && buf[pos] == '#') {
                while (pos < buf.size() && buf[pos] != '\n') pos++;
                continue;
            }
            break;
        }
        long v = -1;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') {
            v = (v < 0 ? 0 : v) * 10 + (buf[pos] - '0');
            pos++;
        }
        return v;
    };
    const bool colour = buf[1] == '6';
    const long w = number(), h = number(), maxv = number();
    pos++;  // the single whitespace byte after maxval
    if (w <= 0 || h <= 0 || maxv != 255) return cv::Mat();
    const std::size_t n = (std::size_t)w * h;
    if (buf.size() < pos + n * (colour ? 3 : 1)) return cv::Mat();
    cv::Mat m((int)h, (int)w, CV_8UC3);
    const unsigned char *src = buf.data() + pos;
    for (std::size_t i = 0; i < n; i++) {
        if (colour) {  // file is RGB
            m.data[3 * i] = src[3 * i + 2];
            m.data[3 * i + 1] = src[3 * i + 1];
            m.data[3 * i + 2] = src[3 * i];
        } else {
            m.data[3 * i] = m.data[3 * i + 1] = m.data[3 * i + 2] = src[i];
        }
    }
    return m;
}

}  // namespace

cv::Mat decode_png(const unsigned char *data, std::size_t size, std::string *error) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (size < 8 || std::memcmp(data, sig, 8) != 0) return fail(error, "not a PNG");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, palette;
    std::size_t pos = 8;
    bool seen_end = false;
    while (pos + 12 <= size && !seen_end) {
        const uint32_t len = be32(data + pos);
        const unsigned char *type = data + pos + 4, *body = data + pos + 8;
        if (pos + 12 + (std::size_t)len > size) return fail(error, "truncated chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return fail(error, "bad IHDR");
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + (std::size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > (1u << 20) || h > (1u << 20)) return fail(error, "bad header");
    if ((uint64_t)w * h > (1ull << 28)) return fail(error, "image too large");
    if (interlace) return fail(error, "interlaced PNG is not supported");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;  // grey
        case 2: channels = 3; break;  // RGB
        case 3: channels = 1; break;  // palette index
        case 4: channels = 2; break;  // grey + alpha
        case 6: channels = 4; break;  // RGBA
        default: return fail(error, "bad colour type");
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) return fail(error, "bad bit depth");
    if (ctype == 3 && palette.size() < 3) return fail(error, "missing palette");
    const std::size_t bits_pp = (std::size_t)channels * depth;
    const std::size_t stride = ((std::size_t)w * bits_pp + 7) / 8;  // bytes per scanline without the filter byte
    const std::size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;         // filter distance in bytes
    std::vector<unsigned char> raw((stride + 1) * (std::size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return fail(error, "inflate failed");
    // undo the scanline filters in place
    std::vector<unsigned char> zero(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        unsigned char *line = raw.data() + (stride + 1) * y + 1;
        const unsigned char *prev = y ? line - (stride + 1) : zero.data();
        const int filter = line[-1];
        for (std::size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? line[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: pred = paeth(a, b, c); break;
                default: return fail(error, "bad filter type");
            }
            line[i] = (unsigned char)(line[i] + pred);
        }
    }
    cv::Mat m((int)h, (int)w, CV_8UC3);
    const int maxv = (1 << depth) - 1;
    for (uint32_t y = 0; y < h; y++) {
        const unsigned char *line = raw.data() + (stride + 1) * y + 1;
        unsigned char *dst = m.data + (std::size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            // sample k of pixel x as an 8-bit value (16-bit: high byte; sub-byte grey: scaled to 0..255)
            auto sample = [&](int k) -> int {
                if (depth == 8) return line[(std::size_t)x * channels + k];
                if (depth == 16) return line[((std::size_t)x * channels + k) * 2];
                const std::size_t bit = (std::size_t)x * depth;
                return (line[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
            };
            int r, g, b;
            if (ctype == 2 || ctype == 6) {
                r = sample(0); g = sample(1); b = sample(2);
            } else if (ctype == 3) {
                const std::size_t idx = (std::size_t)sample(0) * 3;
                if (idx + 2 >= palette.size()) return fail(error, "palette index out of range");
                r = palette[idx]; g = palette[idx + 1]; b = palette[idx + 2];
            } else {
                int v = sample(0);
                if (depth < 8) v = v * 255 / maxv;
                r = g = b = v;
            }
            dst[3 * x] = (unsigned char)b;
            dst[3 * x + 1] = (unsigned char)g;
            dst[3 * x + 2] = (unsigned char)r;
        }
    }
    return m;
}

bool imwrite(const std::string &path, const cv::Mat &bgr) {
    if (bgr.empty() || bgr.type() != CV_8UC3) return false;
    const std::size_t w = (std::size_t)bgr.cols, h = (std::size_t)bgr.rows;
    std::vector<unsigned char> raw((w * 3 + 1) * h);
    for (std::size_t y = 0; y < h; y++) {
        unsigned char *line = raw.data() + (w * 3 + 1) * y;
        const unsigned char *src = bgr.ptr<unsigned char>((int)y);
        *line++ = 0;  // filter type: none
        for (std::size_t x = 0; x < w; x++) {
            line[3 * x] = src[3 * x + 2];
            line[3 * x + 1] = src[3 * x + 1];
            line[3 * x + 2] = src[3 * x];
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK) return false;
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    auto put32 = [](unsigned char *p, uint32_t v) { p[0] = (unsigned char)(v >> 24), p[1] = (unsigned char)(v >> 16), p[2] = (unsigned char)(v >> 8), p[3] = (unsigned char)v; };
    auto chunk = [&](const char *tag, const unsigned char *body, std::size_t n) {
        unsigned char head[8], tail[4];
        put32(head, (uint32_t)n);
        std::memcpy(head + 4, tag, 4);
        uLong crc = crc32(0L, head + 4, 4);
        if (n) crc = crc32(crc, body, (uInt)n);
        put32(tail, (uint32_t)crc);
        f.write(reinterpret_cast<const char *>(head), 8);
        if (n) f.write(reinterpret_cast<const char *>(body), (std::streamsize)n);
        f.write(reinterpret_cast<const char *>(tail), 4);
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    f.write(reinterpret_cast<const char *>(sig), 8);
    unsigned char ihdr[13];
    put32(ihdr, (uint32_t)w);
    put32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8, ihdr[9] = 2, ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;  // 8-bit RGB, no interlace
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), zlen);
    chunk("IEND", nullptr, 0);
    return (bool)f;
}

cv::Mat imread(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return cv::Mat();
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.size() >= 8 && buf[0] == 0x89 && buf[1] == 'P') return decode_png(buf.data(), buf.size());
    if (buf.size() >= 4 && buf[0] == 0xFF && buf[1] == 0xD8) return decode_jpeg(buf.data(), buf.size());
    if (buf.size() >= 7 && buf[0] == 'P' && (buf[1] == '5' || buf[1] == '6')) return read_pnm(buf);
    return cv::Mat();
}


// ---- overlays -----------------------------------------------------------------

namespace {

struct Canvas {
    cv::Mat &m;
    void put(int x, int y, const unsigned char c[3]) {
        if (x < 0 || y < 0 || x >= m.cols || y >= m.rows) return;
        unsigned char *p = m.ptr<unsigned char>(y) + 3 * x;
        p[0] = c[0];
        p[1] = c[1];
        p[2] = c[2];
    }
    void circle(int cx, int cy, int r, const unsigned char c[3]) {  // midpoint circle, one pixel wide
        if (r <= 0) {
            put(cx, cy, c);
            return;
        }
        int x = r, y = 0, err = 1 - r;
        while (x >= y) {
            const int px[8] = {x, y, -y, -x, -x, -y, y, x}, py[8] = {y, x, x, y, -y, -x, -x, -y};
            for (int k = 0; k < 8; k++) put(cx + px[k], cy + py[k], c);
            y++;
            if (err < 0) {
                err += 2 * y + 1;
            } else {
                x--;
                err += 2 * (y - x) + 1;
            }
        }
    }
    void line(int x0, int y0, int x1, int y1, const unsigned char c[3]) {  // Bresenham
        const int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1, dy = -std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
        // clip hopeless segments early: both ends far outside on the same side
        if ((x0 < 0 && x1 < 0) || (y0 < 0 && y1 < 0) || (x0 >= m.cols && x1 >= m.cols) || (y0 >= m.rows && y1 >= m.rows)) return;
        int err = dx + dy;
        for (long guard = 0; guard < 4L * (dx - dy + 1); guard++) {
            put(x0, y0, c);
            if (x0 == x1 && y0 == y1) break;
            const int e2 = 2 * err;
            if (e2 >= dy) {
                err += dy;
                x0 += sx;
            }
            if (e2 <= dx) {
                err += dx;
                y0 += sy;
            }
        }
    }
};

// one colour per item: a fixed multiplicative sequence, bright enough to show on dark and light ground
void colour_of(unsigned index, unsigned char c[3]) {
    unsigned s = (index + 1u) * 2654435761u;
    for (int k = 0; k < 3; k++) {
        s ^= s >> 15;
        s *= 2246822519u;
        c[k] = (unsigned char)(64 + (s >> 24) % 192);
    }
}

cv::Mat to_bgr(const cv::Mat &image) {
    cv::Mat out(image.rows, image.cols, CV_8UC3);
    for (int y = 0; y < image.rows; y++) {
        const unsigned char *src = image.ptr<unsigned char>(y);
        unsigned char *dst = out.ptr<unsigned char>(y);
        if (image.channels() == 3) {
            std::memcpy(dst, src, (size_t)image.cols * 3);
        } else {
            for (int x = 0; x < image.cols; x++) dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = src[x];
        }
    }
    return out;
}

}  // namespace

cv::Mat draw_keypoints(const cv::Mat &image, const std::vector<cv::KeyPoint> &keypoints) {
    if (image.empty() || image.depth() != CV_8U || (image.channels() != 1 && image.channels() != 3)) return cv::Mat();
    cv::Mat out = to_bgr(image);
    Canvas cv_{out};
    for (size_t i = 0; i < keypoints.size(); i++) {
        const cv::KeyPoint &k = keypoints[i];
        unsigned char c[3];
        colour_of((unsigned)i, c);
        const int cx = (int)std::lrint(k.pt.x), cy = (int)std::lrint(k.pt.y), r = (int)std::lrint(k.size / 2);
        cv_.circle(cx, cy, r, c);
        if (k.angle != -1) {
            const double a = k.angle * 3.14159265358979323846 / 180.0;
            cv_.line(cx, cy, cx + (int)std::lrint(std::cos(a) * r), cy + (int)std::lrint(std::sin(a) * r), c);
        }
    }
    return out;
}

cv::Mat draw_matches(const cv::Mat &a, const std::vector<cv::KeyPoint> &ka, const cv::Mat &b,
                     const std::vector<cv::KeyPoint> &kb, const std::vector<std::pair<int, int>> &pairs) {
    if (a.empty() || b.empty() || a.type() != CV_8UC3 || b.type() != CV_8UC3) return cv::Mat();
    const int rows = std::max(a.rows, b.rows);
    cv::Mat out(rows, a.cols + b.cols, CV_8UC3);
    for (int y = 0; y < rows; y++) {
        unsigned char *dst = out.ptr<unsigned char>(y);
        std::memset(dst, 0, (size_t)out.cols * 3);
        if (y < a.rows) std::memcpy(dst, a.ptr<unsigned char>(y), (size_t)a.cols * 3);
        if (y < b.rows) std::memcpy(dst + (size_t)a.cols * 3, b.ptr<unsigned char>(y), (size_t)b.cols * 3);
    }
    Canvas cv_{out};
    for (size_t i = 0; i < pairs.size(); i++) {
        if (pairs[i].first < 0 || pairs[i].first >= (int)ka.size() || pairs[i].second < 0 || pairs[i].second >= (int)kb.size()) continue;
        unsigned char c[3];
        colour_of((unsigned)i, c);
        const cv::KeyPoint &p = ka[(size_t)pairs[i].first], &q = kb[(size_t)pairs[i].second];
        const int x0 = (int)std::lrint(p.pt.x), y0 = (int)std::lrint(p.pt.y);
        const int x1 = a.cols + (int)std::lrint(q.pt.x), y1 = (int)std::lrint(q.pt.y);
        cv_.circle(x0, y0, 3, c);
        cv_.circle(x1, y1, 3, c);
        cv_.line(x0, y0, x1, y1, c);
    }
    return out;
}

}  // namespace sift_io
