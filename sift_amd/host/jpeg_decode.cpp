// JPEG -> 8-bit BGR, with the pixel values libjpeg-turbo produces at its default
// settings (the decoder behind cv::imread for the .jpg/.jpeg files the
// reference is run on: res/*.jpg, main.cpp:30, :41, :62): Huffman baseline and
// progressive, 8-bit samples, grey or three components; the exact integer
// "islow" inverse DCT; "fancy" (triangle-filter) upsampling of 2:1 horizontally
// and 2:1 / 2:2 subsampled chroma, sample replication for other ratios; the
// 16-bit fixed-point YCbCr -> RGB tables; restart intervals; Adobe transform
// flag; the EXIF orientation applied as cv::imread does.  Everything is integer
// arithmetic, so the result is defined bit for bit.
//
// Not supported (empty Mat): arithmetic coding, lossless and hierarchical
// modes, 12-bit samples, four-component (CMYK/YCCK) files.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace sift_io {

namespace {

const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool defined = false;
    int maxcode[18];  // largest code of each length, -1 if none
    int valptr[17];   // index of the first symbol of each length
    int mincode[17];
    unsigned char vals[256];
    short fast[512];  // 9-bit lookahead: (length << 8) | symbol, 0 if longer

    bool build(const unsigned char counts[16], const unsigned char *symbols, int nsym) {
        std::memcpy(vals, symbols, (size_t)nsym);
        int code = 0, k = 0;
        std::memset(fast, 0, sizeof fast);
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
                if (l <= 9) {
                    const int first = code << (9 - l);
                    for (int f = 0; f < (1 << (9 - l)); f++) fast[first + f] = (short)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = counts[l - 1] ? code - 1 : -1;
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
        return true;
    }
};

// Entropy-coded segment reader: 0xFF00 unstuffed; at a marker it stops and feeds zero bits.
struct BitReader {
    const unsigned char *p = nullptr, *end = nullptr;
    uint64_t acc = 0;
    int nbits = 0;

    void fill() {
        while (nbits <= 48) {
            unsigned b = 0;
            if (p < end) {
                if (*p == 0xFF) {
                    if (p + 1 < end && p[1] == 0) {
                        b = 0xFF;
                        p += 2;
                    }  // else: a marker; stay on it
                } else {
                    b = *p++;
                }
            }
            acc |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    }
    int peek(int n) {
        if (nbits < n) fill();
        return (int)(acc >> (64 - n));
    }
    void skip(int n) {
        acc <<= n;
        nbits -= n;
    }
    int get(int n) {
        if (n == 0) return 0;
        const int v = peek(n);
        skip(n);
        return v;
    }
    void reset() {
        acc = 0;
        nbits = 0;
    }
    int decode(const HuffTable &t) {
        const int look = peek(9);
        const int f = t.fast[look];
        if (f) {
            skip(f >> 8);
            return f & 255;
        }
        int code = peek(16), l = 10;
        for (; l <= 16; l++)
            if ((code >> (16 - l)) <= t.maxcode[l]) break;
        if (l > 16) {
            skip(16);
            return 0;  // corrupt data: libjpeg substitutes a zero symbol
        }
        skip(l);
        return t.vals[(t.valptr[l] + (code >> (16 - l)) - t.mincode[l]) & 255];
    }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int width = 0, height = 0;  // samples of the downsampled component that belong to the image
    int bw = 0, bh = 0;         // blocks allocated (whole MCUs)
    int dc_tbl = 0, ac_tbl = 0;
    int pred = 0;
    std::vector<short> coef;           // bw*bh blocks of 64, natural order
    std::vector<unsigned char> plane;  // (bh*8) x (bw*8) samples after the inverse DCT
    short *block(int by, int bx) { return coef.data() + ((size_t)by * bw + bx) * 64; }
};

// jidctint.c (jpeg_idct_islow): 13-bit constants, two passes, results through the range-limit table.
void idct_islow(const short *in, const unsigned short *q, unsigned char *out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                   F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    long ws[64];
    auto pass = [&](long i0, long i1, long i2, long i3, long i4, long i5, long i6, long i7, long o[8]) {
        long z1 = (i2 + i6) * F0_541;
        long tmp2 = z1 + i6 * (-F1_847), tmp3 = z1 + i2 * F0_765;
        long tmp0 = (i0 + i4) * (1L << CB), tmp1 = (i0 - i4) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = i7, tmp1 = i5, tmp2 = i3, tmp3 = i1;
        z1 = tmp0 + tmp3;
        long z2 = tmp1 + tmp2, z3 = tmp0 + tmp2, z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298, tmp1 *= F2_053, tmp2 *= F3_072, tmp3 *= F1_501;
        z1 *= -F0_899, z2 *= -F2_562, z3 *= -F1_961, z4 *= -F0_390;
        z3 += z5, z4 += z5;
        tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
        o[0] = tmp10 + tmp3, o[7] = tmp10 - tmp3, o[1] = tmp11 + tmp2, o[6] = tmp11 - tmp2;
        o[2] = tmp12 + tmp1, o[5] = tmp12 - tmp1, o[3] = tmp13 + tmp0, o[4] = tmp13 - tmp0;
    };
    for (int c = 0; c < 8; c++) {
        long o[8];
        pass((long)in[c] * q[c], (long)in[8 + c] * q[8 + c], (long)in[16 + c] * q[16 + c], (long)in[24 + c] * q[24 + c],
             (long)in[32 + c] * q[32 + c], (long)in[40 + c] * q[40 + c], (long)in[48 + c] * q[48 + c],
             (long)in[56 + c] * q[56 + c], o);
        for (int r = 0; r < 8; r++) ws[8 * r + c] = (o[r] + (1L << (CB - P1 - 1))) >> (CB - P1);
    }
    for (int r = 0; r < 8; r++) {
        const long *w = ws + 8 * r;
        long o[8];
        pass(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], o);
        for (int c = 0; c < 8; c++) {
            const int idx = (int)((o[c] + (1L << (CB + P1 + 2))) >> (CB + P1 + 3)) & 1023;
            // the post-IDCT range-limit table: +128, clamped; its upper half covers the wrapped negatives
            out[r * stride + c] = (unsigned char)(idx < 128 ? idx + 128 : idx < 512 ? 255 : idx < 896 ? 0 : idx - 896);
        }
    }
}

struct Decoder {
    Decoder(const unsigned char *d, size_t n, std::string *e) : data(d), size(n), error(e) { std::memset(quant, 0, sizeof quant); }
    const unsigned char *data;
    size_t size;
    std::string *error;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    bool jfif = false, adobe = false;
    int adobe_transform = 0, orientation = 1, restart_interval = 0;
    Component comp[3];
    unsigned short quant[4][64];
    bool quant_defined[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];

    bool fail(const char *why) {
        if (error) *error = why;
        return false;
    }

    // ---- markers ----------------------------------------------------------------------
    bool parse() {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG");
        size_t pos = 2;
        while (pos + 4 <= size) {
            if (data[pos] != 0xFF) {
                pos++;
                continue;
            }
            const int m = data[pos + 1];
            if (m == 0xFF) {
                pos++;
                continue;
            }
            if (m == 0xD9) return have_frame ? true : fail("no frame");
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD8)) {
                pos += 2;
                continue;
            }
            const size_t len = (size_t)data[pos + 2] << 8 | data[pos + 3];
            if (len < 2 || pos + 2 + len > size) return fail("truncated segment");
            const unsigned char *seg = data + pos + 4;
            const size_t n = len - 2;
            bool ok = true;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) ok = frame(seg, n, m == 0xC2);
            else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) return fail("unsupported JPEG process");
            else if (m == 0xC4) ok = huffman(seg, n);
            else if (m == 0xDB) ok = quantisation(seg, n);
            else if (m == 0xDD && n >= 2) restart_interval = seg[0] << 8 | seg[1];
            else if (m == 0xE0 && n >= 5 && !std::memcmp(seg, "JFIF", 5)) jfif = true;
            else if (m == 0xE1) exif(seg, n);
            else if (m == 0xEE && n >= 12 && !std::memcmp(seg, "Adobe", 5)) adobe = true, adobe_transform = seg[11];
            if (!ok) return false;
            pos += 2 + len;
            if (m == 0xDA) {
                if (!scan(seg, n, pos)) return false;
            }
        }
        return have_frame ? true : fail("no frame");  // a file cut before EOI: what was decoded is returned
    }

    bool frame(const unsigned char *s, size_t n, bool prog) {
        if (have_frame) return fail("second frame header");
        if (n < 6 || s[0] != 8) return fail("only 8-bit samples are supported");
        height = s[1] << 8 | s[2];
        width = s[3] << 8 | s[4];
        ncomp = s[5];
        if (width <= 0 || height <= 0) return fail("bad dimensions");
        if ((long long)width * height > (1LL << 28)) return fail("image too large");
        if (ncomp != 1 && ncomp != 3) return fail("only grey and three-component files are supported");
        if (n < 6 + 3 * (size_t)ncomp) return fail("truncated frame header");
        progressive = prog;
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.id = s[6 + 3 * i];
            c.h = s[7 + 3 * i] >> 4;
            c.v = s[7 + 3 * i] & 15;
            c.tq = s[8 + 3 * i] & 3;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("bad sampling factor");
            hmax = std::max(hmax, c.h);
            vmax = std::max(vmax, c.v);
        }
        if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // a single component is never interleaved
        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.width = (width * c.h + hmax - 1) / hmax;
            c.height = (height * c.v + vmax - 1) / vmax;
            c.bw = mcus_x * c.h;
            c.bh = mcus_y * c.v;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        have_frame = true;
        return true;
    }

    bool huffman(const unsigned char *s, size_t n) {
        while (n >= 17) {
            const int tc = s[0] >> 4, th = s[0] & 15;
            int total = 0;
            for (int i = 0; i < 16; i++) total += s[1 + i];
            if (tc > 1 || th > 3 || total > 256 || n < 17 + (size_t)total) return fail("bad Huffman table");
            if (!(tc ? ac : dc)[th].build(s + 1, s + 17, total)) return fail("bad Huffman table");
            s += 17 + total;
            n -= 17 + (size_t)total;
        }
        return true;
    }

    bool quantisation(const unsigned char *s, size_t n) {
        while (n >= 1) {
            const int prec = s[0] >> 4, tq = s[0] & 15;
            const size_t need = 1 + (prec ? 128 : 64);
            if (tq > 3 || n < need) return fail("bad quantisation table");
            for (int i = 0; i < 64; i++)
                quant[tq][kZigzag[i]] = prec ? (unsigned short)(s[1 + 2 * i] << 8 | s[2 + 2 * i]) : s[1 + i];
            quant_defined[tq] = true;
            s += need;
            n -= need;
        }
        return true;
    }

    // EXIF orientation (tag 0x0112 of IFD0)
    void exif(const unsigned char *s, size_t n) {
        if (n < 14 || std::memcmp(s, "Exif\0\0", 6)) return;
        const unsigned char *t = s + 6;
        const size_t tn = n - 6;
        const bool le = t[0] == 'I' && t[1] == 'I';
        if (!le && !(t[0] == 'M' && t[1] == 'M')) return;
        auto u16 = [&](size_t o) -> unsigned { return le ? t[o] | t[o + 1] << 8 : t[o] << 8 | t[o + 1]; };
        auto u32 = [&](size_t o) -> size_t {
            return le ? (size_t)t[o] | (size_t)t[o + 1] << 8 | (size_t)t[o + 2] << 16 | (size_t)t[o + 3] << 24
                      : (size_t)t[o] << 24 | (size_t)t[o + 1] << 16 | (size_t)t[o + 2] << 8 | (size_t)t[o + 3];
        };
        const size_t ifd = u32(4);
        if (ifd + 2 > tn) return;
        const unsigned entries = u16(ifd);
        for (unsigned e = 0; e < entries; e++) {
            const size_t o = ifd + 2 + 12 * (size_t)e;
            if (o + 12 > tn) return;
            if (u16(o) == 0x0112 && u16(o + 2) == 3) {
                const unsigned v = u16(o + 8);
                if (v >= 1 && v <= 8) orientation = (int)v;
                return;
            }
        }
    }

    // ---- one scan ---------------------------------------------------------------------
    bool scan(const unsigned char *s, size_t n, size_t &pos) {
        if (!have_frame) return fail("scan before frame");
        if (n < 1) return fail("bad scan header");
        const int ns = s[0];
        if (ns < 1 || ns > ncomp || n < 1 + 2 * (size_t)ns + 3) return fail("bad scan header");
        Component *sc[3];
        for (int i = 0; i < ns; i++) {
            sc[i] = nullptr;
            for (int k = 0; k < ncomp; k++)
                if (comp[k].id == s[1 + 2 * i]) sc[i] = &comp[k];
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->dc_tbl = s[2 + 2 * i] >> 4;
            sc[i]->ac_tbl = s[2 + 2 * i] & 15;
            if (sc[i]->dc_tbl > 3 || sc[i]->ac_tbl > 3) return fail("bad table selector");
        }
        int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns];
        const int Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
        if (!progressive) Ss = 0, Se = 63;
        if (Ss > Se || Se > 63 || (progressive && Ss == 0 && Se != 0) || (progressive && Ss > 0 && ns != 1))
            return fail("bad spectral selection");
        for (int i = 0; i < ns; i++) {
            if (Ss == 0 && !(progressive && Ah) && !dc[sc[i]->dc_tbl].defined) return fail("missing DC table");
            if (Se > 0 && !ac[sc[i]->ac_tbl].defined) return fail("missing AC table");
        }

        BitReader br;
        br.p = data + pos;
        br.end = data + size;
        int eobrun = 0;
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;

        auto decode_block = [&](Component &c, short *b) {
            if (!progressive) {
                const int t = std::min(br.decode(dc[c.dc_tbl]), 16);  // a larger category only occurs in corrupt data
                c.pred += t ? extend(br.get(t), t) : 0;
                b[0] = (short)c.pred;
                const HuffTable &h = ac[c.ac_tbl];
                for (int k = 1; k < 64; k++) {
                    const int rs = br.decode(h), r = rs >> 4, sz = rs & 15;
                    if (sz) {
                        k += r;
                        if (k > 63) break;
                        b[kZigzag[k]] = (short)extend(br.get(sz), sz);
                    } else {
                        if (r != 15) break;
                        k += 15;
                    }
                }
            } else if (Ss == 0) {
                if (Ah == 0) {
                    const int t = std::min(br.decode(dc[c.dc_tbl]), 16);  // a larger category only occurs in corrupt data
                    c.pred += t ? extend(br.get(t), t) : 0;
                    b[0] = (short)(c.pred * (1 << Al));
                } else if (br.get(1)) {
                    b[0] |= (short)(1 << Al);
                }
            } else if (Ah == 0) {  // AC, first pass of these bits
                if (eobrun > 0) {
                    eobrun--;
                    return;
                }
                const HuffTable &h = ac[c.ac_tbl];
                for (int k = Ss; k <= Se; k++) {
                    const int rs = br.decode(h), r = rs >> 4, sz = rs & 15;
                    if (sz) {
                        k += r;
                        if (k > 63) break;
                        b[kZigzag[k]] = (short)(extend(br.get(sz), sz) * (1 << Al));
                    } else if (r == 15) {
                        k += 15;
                    } else {
                        eobrun = 1 << r;
                        if (r) eobrun += br.get(r);
                        eobrun--;
                        break;
                    }
                }
            } else {  // AC refinement
                const int p1 = 1 << Al, m1 = -(1 << Al);
                int k = Ss;
                auto correct = [&](short &cf) {
                    if (br.get(1) && (cf & p1) == 0) cf = (short)(cf >= 0 ? cf + p1 : cf + m1);
                };
                if (eobrun == 0) {
                    const HuffTable &h = ac[c.ac_tbl];
                    for (; k <= Se; k++) {
                        const int rs = br.decode(h);
                        int r = rs >> 4, sz = rs & 15;
                        if (sz) {
                            sz = br.get(1) ? p1 : m1;
                        } else if (r != 15) {
                            eobrun = 1 << r;
                            if (r) eobrun += br.get(r);
                            break;
                        }
                        do {
                            short &cf = b[kZigzag[k]];
                            if (cf != 0) {
                                correct(cf);
                            } else if (--r < 0) {
                                break;
                            }
                            k++;
                        } while (k <= Se);
                        if (sz && k <= 63) b[kZigzag[k]] = (short)sz;
                    }
                }
                if (eobrun > 0) {
                    for (; k <= Se; k++) {
                        short &cf = b[kZigzag[k]];
                        if (cf != 0) correct(cf);
                    }
                    eobrun--;
                }
            }
        };

        auto restart = [&]() {
            br.reset();
            // the reader stopped in front of the marker; skip fill bytes and the RSTn itself
            while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) {
                if (br.p[0] == 0xFF && br.p[1] != 0 && br.p[1] != 0xFF) return;  // some other marker: leave it to parse()
                br.p++;
            }
            if (br.p + 1 < br.end) br.p += 2;
            for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
            eobrun = 0;
        };

        int to_go = restart_interval;
        if (ns == 1) {
            Component &c = *sc[0];
            const int nbx = (c.width + 7) / 8, nby = (c.height + 7) / 8;
            for (int by = 0; by < nby; by++)
                for (int bx = 0; bx < nbx; bx++) {
                    if (restart_interval && to_go == 0) {
                        restart();
                        to_go = restart_interval;
                    }
                    decode_block(c, c.block(by, bx));
                    to_go--;
                }
        } else {
            const int mcus_x = comp[0].bw / comp[0].h, mcus_y = comp[0].bh / comp[0].v;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart_interval && to_go == 0) {
                        restart();
                        to_go = restart_interval;
                    }
                    for (int i = 0; i < ns; i++)
                        for (int y = 0; y < sc[i]->v; y++)
                            for (int x = 0; x < sc[i]->h; x++) decode_block(*sc[i], sc[i]->block(my * sc[i]->v + y, mx * sc[i]->h + x));
                    to_go--;
                }
        }
        // continue parsing at the marker the reader stopped in front of
        const unsigned char *q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        pos = (size_t)(q - data);
        return true;
    }

    // ---- reconstruction ---------------------------------------------------------------
    bool reconstruct(cv::Mat &out) {
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            if (!quant_defined[c.tq]) return fail("missing quantisation table");
            const int stride = c.bw * 8;
            c.plane.assign((size_t)stride * c.bh * 8, 0);
            for (int by = 0; by < c.bh; by++)
                for (int bx = 0; bx < c.bw; bx++)
                    idct_islow(c.block(by, bx), quant[c.tq], c.plane.data() + ((size_t)by * 8) * stride + bx * 8, stride);
        }
        // full-resolution rows of each component
        std::vector<unsigned char> full[3];
        for (int i = 0; i < ncomp; i++) upsample(comp[i], full[i]);
        cv::Mat m(height, width, CV_8UC3);
        bool ycc = true;
        if (ncomp == 3) {
            if (adobe) ycc = adobe_transform != 0;
            else if (!jfif && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
        }
        // jdcolor.c: 16-bit fixed point tables; green takes the sum of its two halves, then one shift
        int cr_r[256], cb_b[256];
        long cr_g[256], cb_g[256];
        for (int i = 0; i < 256; i++) {
            const long x = i - 128;
            cr_r[i] = (int)((91881L * x + 32768) >> 16);   // FIX(1.40200)
            cb_b[i] = (int)((116130L * x + 32768) >> 16);  // FIX(1.77200)
            cr_g[i] = -46802L * x;                         // FIX(0.71414)
            cb_g[i] = -22554L * x + 32768;                 // FIX(0.34414)
        }
        auto clamp = [](int v) { return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); };
        for (int y = 0; y < height; y++) {
            unsigned char *dst = m.data + (size_t)y * width * 3;
            const size_t row = (size_t)y * width;
            for (int x = 0; x < width; x++) {
                int r, g, b;
                if (ncomp == 1) {
                    r = g = b = full[0][row + x];
                } else if (ycc) {
                    const int Y = full[0][row + x], cb = full[1][row + x], cr = full[2][row + x];
                    r = clamp(Y + cr_r[cr]);
                    g = clamp(Y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
                    b = clamp(Y + cb_b[cb]);
                } else {
                    r = full[0][row + x], g = full[1][row + x], b = full[2][row + x];
                }
                dst[3 * x] = (unsigned char)b;
                dst[3 * x + 1] = (unsigned char)g;
                dst[3 * x + 2] = (unsigned char)r;
            }
        }
        out = oriented(m);
        return true;
    }

    // jdsample.c: component -> width x height samples.
    void upsample(const Component &c, std::vector<unsigned char> &full) {
        full.assign((size_t)width * height, 0);
        const int stride = c.bw * 8;
        const int hx = hmax / c.h, vx = vmax / c.v;
        const bool integral = hmax % c.h == 0 && vmax % c.v == 0;
        const unsigned char *src = c.plane.data();
        const int cw = c.width, ch = c.height;
        if (hx == 1 && vx == 1) {
            for (int y = 0; y < height; y++) std::memcpy(&full[(size_t)y * width], src + (size_t)y * stride, (size_t)width);
        } else if (integral && hx == 2 && vx == 1 && cw > 2) {  // h2v1_fancy_upsample
            std::vector<unsigned char> line((size_t)2 * cw);
            for (int y = 0; y < height; y++) {
                const unsigned char *in = src + (size_t)y * stride;
                line[0] = in[0];
                line[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
                for (int x = 1; x < cw - 1; x++) {
                    line[2 * x] = (unsigned char)((in[x] * 3 + in[x - 1] + 1) >> 2);
                    line[2 * x + 1] = (unsigned char)((in[x] * 3 + in[x + 1] + 2) >> 2);
                }
                line[2 * cw - 2] = (unsigned char)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
                line[2 * cw - 1] = in[cw - 1];
                std::memcpy(&full[(size_t)y * width], line.data(), (size_t)width);
            }
        } else if (integral && hx == 2 && vx == 2 && cw > 2) {  // h2v2_fancy_upsample
            std::vector<unsigned char> line((size_t)2 * cw);
            for (int y = 0; y < height; y++) {
                const int cy = y >> 1;
                // the nearer row weighs 3, the farther 1; rows beyond the component repeat its edge rows
                const int oy = (y & 1) ? std::min(cy + 1, ch - 1) : std::max(cy - 1, 0);
                const unsigned char *in0 = src + (size_t)cy * stride, *in1 = src + (size_t)oy * stride;
                int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                line[0] = (unsigned char)((thiscol * 4 + 8) >> 4);
                line[1] = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol;
                thiscol = nextcol;
                for (int x = 1; x < cw - 1; x++) {
                    nextcol = in0[x + 1] * 3 + in1[x + 1];
                    line[2 * x] = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
                    line[2 * x + 1] = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol;
                    thiscol = nextcol;
                }
                line[2 * cw - 2] = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
                line[2 * cw - 1] = (unsigned char)((thiscol * 4 + 7) >> 4);
                std::memcpy(&full[(size_t)y * width], line.data(), (size_t)width);
            }
        } else if (integral && hx == 1 && vx == 2) {  // h1v2_fancy_upsample (libjpeg-turbo 1.5 and later)
            for (int y = 0; y < height; y++) {
                const int cy = y >> 1, bias = (y & 1) ? 2 : 1;
                const int oy = (y & 1) ? std::min(cy + 1, ch - 1) : std::max(cy - 1, 0);
                const unsigned char *in0 = src + (size_t)cy * stride, *in1 = src + (size_t)oy * stride;
                for (int x = 0; x < width; x++) full[(size_t)y * width + x] = (unsigned char)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        } else {  // sample replication (int_upsample and its special cases); non-integral ratios take the nearest sample below
            for (int y = 0; y < height; y++) {
                const int cy = std::min(integral ? y / vx : y * c.v / vmax, ch - 1);
                const unsigned char *in = src + (size_t)cy * stride;
                for (int x = 0; x < width; x++) full[(size_t)y * width + x] = in[std::min(integral ? x / hx : x * c.h / hmax, cw - 1)];
            }
        }
    }

    // The eight EXIF orientations as cv::imread applies them (transpose, then flips).
    cv::Mat oriented(const cv::Mat &m) const {
        if (orientation <= 1 || orientation > 8) return m;
        const bool transpose = orientation >= 5;
        const int W = m.cols, H = m.rows, ow = transpose ? H : W, oh = transpose ? W : H;
        bool flip_x, flip_y;  // applied after the optional transpose
        switch (orientation) {
            case 2: flip_x = true, flip_y = false; break;
            case 3: flip_x = true, flip_y = true; break;
            case 4: flip_x = false, flip_y = true; break;
            case 5: flip_x = false, flip_y = false; break;
            case 6: flip_x = true, flip_y = false; break;
            case 7: flip_x = true, flip_y = true; break;
            default: flip_x = false, flip_y = true; break;  // 8
        }
        cv::Mat o(oh, ow, CV_8UC3);
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++) {
                const int tx = flip_x ? ow - 1 - x : x, ty = flip_y ? oh - 1 - y : y;  // position before the flips
                const int sx = transpose ? ty : tx, sy = transpose ? tx : ty;          // position before the transpose
                std::memcpy(o.data + ((size_t)y * ow + x) * 3, m.data + ((size_t)sy * W + sx) * 3, 3);
            }
        return o;
    }
};

}  // namespace

cv::Mat decode_jpeg(const unsigned char *data, std::size_t size, std::string *error) {
    Decoder d(data, size, error);
    cv::Mat out;
    if (!d.parse() || !d.reconstruct(out)) return cv::Mat();
    return out;
}

}  // namespace sift_io
