// Gaussian scale-space pyramid.
//
// Replaces gen_gaussian_images (sift.cpp:132-156) and the base-image part of
// the constructor (sift.cpp:40-42).  Every blur is one launch that reads its
// input level once and writes the blurred level and, for level 3, the
// nearest-neighbour half-size base of the next octave (sift.cpp:152-153) in
// the same pass.  The difference-of-Gaussians images of gen_dog_images
// (sift.cpp:161-177) are never stored: their only readers, the extrema scan
// and the localisation, subtract two Gaussian levels in registers, which is
// the same IEEE operation on the same operands.
//
// Marching structure: a workgroup owns a strip of 128 (or 64) columns and walks
// down it 16 (or 32) rows at a time.  Each step stages the new input rows (plus
// the horizontal halo) in an LDS buffer, row-filters them into a ring
// of row-filtered rows (also LDS) and then column-filters the output rows whose
// 2R+1 supports are now complete.  Every input row is read from HBM and
// row-filtered once per strip (no vertical halo recompute); the next step's
// input is prefetched into registers while the current step computes; every
// thread has exactly one row item and one column item per step.  Rows above/below the image are the reflected rows
// (BORDER_REFLECT_101), re-filtered.
//
// Arithmetic order is fixed (and contraction is off for this file) so that
// results are bit-identical to a scalar CPU evaluation of the same filter:
// rows accumulate taps left to right, s = c0*v0; s += ct*vt; columns start at
// the centre tap and add symmetric pairs, s = cR*vR; s += c(R+t)*(v(R+t)+v(R-t)).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace sa {

namespace {

// Strip geometry.  Filters up to 17 taps use 128-column strips walked 16 rows at a time
// (1 KB contiguous per row: fewer, longer bursts and half the column halo).  The 21- and 27-tap
// filters are bound by their arithmetic and need three workgroups per CU to hide it: 64 columns
// x 32 rows keeps their LDS footprint under a third of the CU's 160 KB (the 21-tap filter's
// 128 x 16 tile is 56 KB and fits only twice; 3.14 -> 2.90 ms per 64 1080p frames).

// Threads per workgroup: 256 (8 consecutive outputs per thread in either pass) for the short
// filters, 512 (4 outputs per thread, half the registers, twice the waves per SIMD) for the
// long ones, whose loops are bound by fp64 issue and need the extra waves to stay busy.
template <int R>
struct MarchGeom {
    static constexpr int TW = R <= 8 ? 128 : 64;   // strip width (outputs)
    static constexpr int MCH = R <= 8 ? 16 : 32;   // rows per step
    static constexpr int NTHREADS = 512;
    static constexpr int CX = TW / (NTHREADS / MCH);   // consecutive outputs per thread in the row pass
    static constexpr int CY = MCH / (NTHREADS / TW);   // consecutive outputs per thread in the column pass
    static constexpr int COLS = TW + 2 * R;
    static constexpr int PIN = COLS | 1;   // odd pitch: lane<->row reads are conflict-free
    static constexpr int RING = MCH + 2 * R;
    static constexpr int PR = TW + 1;      // odd pitch: lane<->row writes are conflict-free
    static constexpr size_t LDS_BYTES = ((size_t)MCH * PIN + (size_t)RING * PR) * sizeof(double);
};

// Source of the blur levels: the previous Gaussian level.
struct LevelSource {
    static constexpr bool kQuads = false;
    const double *in;
    int w;
    __device__ __forceinline__ double operator()(int gy, int gx) const { return in[(size_t)gy * w + gx]; }
};

// Source of the base blur: the 8-bit frame upsampled 2x on the fly.  This is
// resize(INTER_LINEAR) to exactly twice the size (sift.cpp:40): source
// coordinate (d+0.5)/2-0.5 clamped at the edges, i.e. even d = 2k takes
// 0.25*s[k-1] + 0.75*s[k] and odd d = 2k+1 takes 0.75*s[k] + 0.25*s[k+1];
// 8-bit inputs make every product and sum exact in double.
//
// The staging tile of the base blur starts at an odd row and an odd column, so it
// splits into 2x2 output quads (rows 2j+1, 2j+2; columns 2k+1, 2k+2) that all
// interpolate between the same four source pixels a = s[j][k], b = s[j][k+1],
// c = s[j+1][k], d = s[j+1][k+1]: (9a+3b+3c+d)/16 and its three mirror images,
// exact in integers.  `load_quad` only loads (so that the loads of the next step
// stay in flight during this step's filter), `blend` does the arithmetic; quads
// that touch or cross the frame border are recomputed pixel by pixel.
struct UpsampleSource {
    static constexpr bool kQuads = true;
    const uint8_t *src;
    int W, H;
    __device__ __forceinline__ void load_quad(int gy, int gx, int raw[4]) const {
        const int j = (gy - 1) >> 1, k = (gx - 1) >> 1;
        const int j0 = min(max(j, 0), H - 1), j1 = min(max(j + 1, 0), H - 1);
        const int k0 = min(max(k, 0), W - 1), k1 = min(max(k + 1, 0), W - 1);
        raw[0] = src[(unsigned)(j0 * W + k0)];
        raw[1] = src[(unsigned)(j0 * W + k1)];
        raw[2] = src[(unsigned)(j1 * W + k0)];
        raw[3] = src[(unsigned)(j1 * W + k1)];
    }
    __device__ __forceinline__ bool interior(int gy, int gx) const {
        return gy >= 1 && gy + 1 < 2 * H && gx >= 1 && gx + 1 < 2 * W;
    }
    static __device__ __forceinline__ void blend(const int raw[4], double o[4]) {
        const int a = raw[0], b = raw[1], c = raw[2], d = raw[3];
        o[0] = (double)(9 * a + 3 * b + 3 * c + d) * 0.0625;
        o[1] = (double)(3 * a + 9 * b + c + 3 * d) * 0.0625;
        o[2] = (double)(3 * a + b + 9 * c + 3 * d) * 0.0625;
        o[3] = (double)(a + 3 * b + 3 * c + 9 * d) * 0.0625;
    }
    __device__ __forceinline__ double operator()(int gy, int gx) const {
        int sx = (gx - 1) >> 1, sy = (gy - 1) >> 1;  // floor((d-1)/2)
        double fx = (gx & 1) ? 0.25 : 0.75, fy = (gy & 1) ? 0.25 : 0.75;
        if (sx < 0) { sx = 0; fx = 0; }
        if (sx >= W - 1) { sx = W - 1; fx = 0; }
        if (sy < 0) { sy = 0; fy = 0; }
        if (sy >= H - 1) { sy = H - 1; fy = 0; }
        const int sx1 = min(sx + 1, W - 1), sy1 = min(sy + 1, H - 1);
        const double a = src[(size_t)sy * W + sx], b = src[(size_t)sy * W + sx1];
        const double c = src[(size_t)sy1 * W + sx], d = src[(size_t)sy1 * W + sx1];
        const double r0 = a * (1.0 - fx) + b * fx, r1 = c * (1.0 - fx) + d * fx;
        return r0 * (1.0 - fy) + r1 * fy;
    }
};

// Keeps the chunked LDS reads of the filter passes where they are written: the memory clobber stops instruction
// selection from gathering every read of a pass at its top (reads are free to move across a scheduling barrier when
// the instruction order is first laid down), the scheduling barrier stops the machine scheduler from doing the same.
__device__ __forceinline__ void stream_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// grid: (strips, vertical segments, frames).  `nxt` may be null.
template <int R, class Source>
__device__ __forceinline__ void march(const Source source, double *__restrict__ out,
                                      double *__restrict__ nxt, int w, int h, int nw, int nh, int strip_h,
                                      const Taps &taps, double *smem, const TileId tile) {
    using G = MarchGeom<R>;
    constexpr int NTHREADS = G::NTHREADS, CX = G::CX, CY = G::CY, TW = G::TW, MCH = G::MCH;
    double *ring = smem + MCH * G::PIN;
    const int tid = threadIdx.x;
    const int tx0 = tile.x * TW;
    const int ys = tile.y * strip_h;
    const int ye = min(h, ys + strip_h);
    const int nsteps = (ye - ys + 2 * R + MCH - 1) / MCH;

    // This thread's share of a MCH x COLS staging tile.  `fetch` issues the loads of step s into registers,
    // `stage` writes them to the LDS staging tile.
    constexpr bool kQuads = Source::kQuads;
    constexpr int QC = G::COLS / 2, NQUAD = (MCH / 2) * QC;
    // Blur levels: a thread owns one row of the tile and every TPR-th column of it, so that a step's loads and LDS
    // writes are one base address plus compile-time offsets (no per-slot index registers live across the loop).
    constexpr int TPR = NTHREADS / MCH, NS = (G::COLS + TPR - 1) / TPR;
    constexpr int NSLOT = kQuads ? (NQUAD + NTHREADS - 1) / NTHREADS : NS;
    static_assert(!kQuads || ((R & 1) && !(MCH & 1) && !(TW & 1)), "quads need an odd tile origin and even tile sides");
    const int srow = tid / TPR, sc0 = tid % TPR;
    const bool edge_x = tx0 - R < 0 || tx0 + TW + R > w;  // uniform over the workgroup: columns need reflecting
    int pf_row[kQuads ? NSLOT : 1], pf_gx[kQuads ? NSLOT : 1], pf_lds[kQuads ? NSLOT : 1];
    double pf[kQuads ? 1 : NSLOT];
    int raw[kQuads ? NSLOT : 1][4];
    if constexpr (kQuads) {
#pragma unroll
        for (int u = 0; u < NSLOT; u++) {
            const int idx = tid + NTHREADS * u;
            const int r = 2 * (idx / QC), c = 2 * (idx % QC);
            pf_row[u] = idx < NQUAD ? r : -1;
            pf_gx[u] = tx0 - R + c;  // not reflected: border quads are recomputed
            pf_lds[u] = r * G::PIN + c;
        }
    }
    auto fetch = [&](int s) {
        const int v0 = ys - R + MCH * s;
        if constexpr (kQuads) {
#pragma unroll
            for (int u = 0; u < NSLOT; u++)
                if ((u + 1) * NTHREADS <= NQUAD || pf_row[u] >= 0) source.load_quad(v0 + pf_row[u], pf_gx[u], raw[u]);
        } else {
            int gy = v0 + srow;
            if ((unsigned)gy >= (unsigned)h) gy = reflect101(gy, h);  // only the strips at the image edge
            if (!edge_x) {
                const int gx0 = tx0 - R + sc0;
#pragma unroll
                for (int u = 0; u < NS; u++)
                    if ((u + 1) * TPR <= G::COLS || sc0 + TPR * u < G::COLS) pf[u] = source(gy, gx0 + TPR * u);
            } else {
#pragma unroll
                for (int u = 0; u < NS; u++)
                    if ((u + 1) * TPR <= G::COLS || sc0 + TPR * u < G::COLS)
                        pf[u] = source(gy, reflect101(tx0 - R + sc0 + TPR * u, w));
            }
        }
    };
    auto stage = [&](int s, double *s_in) {
        const int v0 = ys - R + MCH * s;
        if constexpr (kQuads) {
#pragma unroll
            for (int u = 0; u < NSLOT; u++) {
                if (pf_row[u] < 0) continue;
                const int gy = v0 + pf_row[u], gx = pf_gx[u];
                double o[4];
                if (source.interior(gy, gx)) {
                    Source::blend(raw[u], o);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = source(reflect101(gy + (i >> 1), h), reflect101(gx + (i & 1), w));
                }
                double *q = s_in + pf_lds[u];
                q[0] = o[0];
                q[1] = o[1];
                q[G::PIN] = o[2];
                q[G::PIN + 1] = o[3];
            }
        } else {
            double *q = s_in + srow * G::PIN + sc0;
#pragma unroll
            for (int u = 0; u < NS; u++)
                if ((u + 1) * TPR <= G::COLS || sc0 + TPR * u < G::COLS) q[TPR * u] = pf[u];
        }
        (void)v0;
    };

    fetch(0);
    for (int s = 0; s < nsteps; s++) {
        // One staging tile is enough: every wave has finished the row pass of the previous step (it lies before
        // that step's second barrier) when the first wave gets here.
        double *s_in = smem;
        stage(s, s_in);
        __syncthreads();  // staging tile complete; previous step's column pass is done with the ring
        if (s + 1 < nsteps) fetch(s + 1);
        {   // row pass: lane <-> row, 8 consecutive outputs per thread
            const int row = tid & (MCH - 1), cx = tid / MCH;
            const double *p = s_in + row * G::PIN + cx * CX;
            double *q = ring + ((MCH * s + row) % G::RING) * G::PR + cx * CX;
            // The window of CX + 2R inputs is streamed through registers ECH elements at a time, the next chunk's
            // LDS reads issued before the current chunk is consumed; an element k adds its term c[k-m]*v[k] to
            // every output m it belongs to, so each output still accumulates its taps left to right.  The
            // scheduling barriers keep the compiler from hoisting every read to the top (which costs 2(CX+2R)
            // registers and a third of the waves).
            constexpr int NEL = CX + 2 * R, ECH = 4, NCH = (NEL + ECH - 1) / ECH;
            double acc[CX];
            double buf[2][ECH];
#pragma unroll
            for (int u = 0; u < ECH; u++) buf[0][u] = p[u];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (c + 1 < NCH) {
#pragma unroll
                    for (int u = 0; u < ECH; u++)
                        if ((c + 1) * ECH + u < NEL) buf[(c + 1) & 1][u] = p[(c + 1) * ECH + u];
                }
                stream_fence();
#pragma unroll
                for (int u = 0; u < ECH; u++) {
                    const int k = c * ECH + u;
                    if (k >= NEL) break;
                    const double vk = buf[c & 1][u];
#pragma unroll
                    for (int m = 0; m < CX; m++) {
                        const int t = k - m;
                        if (t == 0) acc[m] = taps.c[0] * vk;
                        else if (t > 0 && t <= 2 * R) acc[m] += taps.c[t] * vk;
                    }
                }
                stream_fence();
            }
#pragma unroll
            for (int m = 0; m < CX; m++) q[m] = acc[m];
        }
        // column pass: lane <-> column, consecutive output rows per thread.  A wave lies in one
        // row group: its ring slots, centre rows and output rows are scalars.
        const int x = tid & (TW - 1), cy = __builtin_amdgcn_readfirstlane(tid / TW);
        const int rel0 = MCH * s + cy * CY - R;  // strip-relative index of the first output row
        __syncthreads();
        {
            const int y0 = ys - R + rel0;  // image row of the first output row
            const int gx = tx0 + x;
            if (y0 + CY > ys && y0 < ye && gx < w) {
                // Ring row of window element k (0 .. CY+2R-1); the base slot is a scalar.
                const int slot0 = (rel0 - R + 8 * G::RING) % G::RING;
                auto ld = [&](int k) {
                    int sl = slot0 + k;
                    sl = sl >= G::RING ? sl - G::RING : sl;
                    return ring[sl * G::PR + x];
                };
                // Output m is c[R]*v[m+R] plus, for t = 1..R in this order, c[R+t]*(v[m+R+t] + v[m+R-t]).  The
                // upper and lower operands of the CY outputs at step t are two windows of CY elements that slide
                // outwards by one element per step, so the column is streamed through registers too: TCH steps'
                // worth of new elements are read one chunk ahead of their use.
                constexpr int TCH = 2, NTC = (R + TCH - 1) / TCH;
                double acc[CY], hi_w[CY], lo_w[CY];
                {
                    double cen[CY];
#pragma unroll
                    for (int m = 0; m < CY; m++) cen[m] = ld(R + m);
#pragma unroll
                    for (int m = 0; m < CY; m++) acc[m] = taps.c[R] * cen[m];
#pragma unroll
                    for (int m = 0; m + 1 < CY; m++) hi_w[m] = cen[m + 1];
                    hi_w[CY - 1] = ld(R + CY);
#pragma unroll
                    for (int m = 1; m < CY; m++) lo_w[m] = cen[m - 1];
                    lo_w[0] = ld(R - 1);
                }
                double fresh_hi[2][TCH], fresh_lo[2][TCH];
                // after step t the windows need v[R+t+CY] and v[R-t-1] (nothing after the last step)
#pragma unroll
                for (int u = 0; u < TCH; u++) {
                    const int t = 1 + u;
                    if (t < R) {
                        fresh_hi[0][u] = ld(R + t + CY);
                        fresh_lo[0][u] = ld(R - t - 1);
                    }
                }
#pragma unroll
                for (int c = 0; c < NTC; c++) {
                    if (c + 1 < NTC) {
#pragma unroll
                        for (int u = 0; u < TCH; u++) {
                            const int t = 1 + (c + 1) * TCH + u;
                            if (t < R) {
                                fresh_hi[(c + 1) & 1][u] = ld(R + t + CY);
                                fresh_lo[(c + 1) & 1][u] = ld(R - t - 1);
                            }
                        }
                    }
                    stream_fence();
#pragma unroll
                    for (int u = 0; u < TCH; u++) {
                        const int t = 1 + c * TCH + u;
                        if (t > R) break;
#pragma unroll
                        for (int m = 0; m < CY; m++) acc[m] += taps.c[R + t] * (hi_w[m] + lo_w[m]);
                        if (t < R) {
#pragma unroll
                            for (int m = 0; m + 1 < CY; m++) hi_w[m] = hi_w[m + 1];
                            hi_w[CY - 1] = fresh_hi[c & 1][u];
#pragma unroll
                            for (int m = CY - 1; m > 0; m--) lo_w[m] = lo_w[m - 1];
                            lo_w[0] = fresh_lo[c & 1][u];
                        }
                    }
                    // the sums so far are pinned here, otherwise the arithmetic of every chunk sinks below the last
                    // read and all 2R + CY elements are live at once
#pragma unroll
                    for (int m = 0; m < CY; m++) asm volatile("" : "+v"(acc[m]));
                    stream_fence();
                }
#pragma unroll
                for (int m = 0; m < CY; m++) {
                    const int y = y0 + m;
                    if (y >= ys && y < ye) {
                        // the row is wave-uniform: scalar row pointers, one vector offset
                        const size_t orow = (size_t)y * w;
                        const unsigned xoff = (unsigned)gx * 8u;  // byte offset inside the row
                        *reinterpret_cast<double *>(reinterpret_cast<char *>(out + orow) + xoff) = acc[m];
                        if (nxt && !(y & 1) && (y >> 1) < nh) {
                            const int hx = gx >> 1;
                            if (!(gx & 1) && hx < nw)
                                *reinterpret_cast<double *>(reinterpret_cast<char *>(nxt + (size_t)(y >> 1) * nw) + xoff / 2) = acc[m];
                        }
                    }
                }
            }
        }
    }
}

// One blur level of one octave for every frame of the batch.
template <int R>
__global__ __launch_bounds__(MarchGeom<R>::NTHREADS) void k_blur_march(const double *__restrict__ in, double *__restrict__ out,
                                                          double *__restrict__ next, int w, int h, int nw, int nh,
                                                          size_t in_frame_stride, size_t next_frame_stride,
                                                          int strip_h, Taps taps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const TileId tile = xcd_tile_id();
    const size_t f = tile.z;
    const LevelSource source{in + f * in_frame_stride, w};
    march<R>(source, out + f * in_frame_stride, next ? next + f * next_frame_stride : nullptr, w, h, nw, nh, strip_h, taps,
             smem, tile);
}

// gray u8 -> double -> 2x bilinear upsample -> base blur (sift.cpp:30, :40-42).
template <int R>
__global__ __launch_bounds__(MarchGeom<R>::NTHREADS) void k_base_march(const uint8_t *__restrict__ frames, int W, int H,
                                                          double *__restrict__ out, size_t out_frame_stride,
                                                          int strip_h, Taps taps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const TileId tile = xcd_tile_id();
    const size_t f = tile.z;
    const UpsampleSource source{frames + f * (size_t)W * H, W, H};
    march<R>(source, out + f * out_frame_stride, nullptr, 2 * W, 2 * H, 0, 0, strip_h, taps, smem, tile);
}

// strips cut vertically so that the launch has >= ~2048 workgroups
template <int R>
void strip_grid(int w, int h, int nframes, dim3 &grid, int &strip_h) {
    constexpr int TW = MarchGeom<R>::TW, MCH = MarchGeom<R>::MCH;
    const int nx = (w + TW - 1) / TW;
    int ny = (2048 + nx * nframes - 1) / (nx * nframes);
    // At least 32 rows per strip.  Short strips repeat their 2R halo rows, which only a launch that cannot fill
    // the chip anyway (a single frame) gets to: there the shorter walk of each workgroup is the latency of the level.
    const int max_ny = (h + 31) / 32;
    if (ny > max_ny) ny = max_ny;
    if (ny < 1) ny = 1;
    strip_h = (h + ny - 1) / ny;
    strip_h = (strip_h + MCH - 1) / MCH * MCH;
    ny = (h + strip_h - 1) / strip_h;
    grid = dim3(nx, ny, nframes);
}

template <int R>
void launch_blur_r(hipStream_t st, const Pyramid &p, int oct, int lvl, int nframes, const Taps &taps) {
    const int w = p.w[oct], h = p.h[oct];
    const size_t plane = (size_t)w * h;
    const double *in = p.gauss[oct] + (size_t)(lvl - 1) * plane;
    double *out = p.gauss[oct] + (size_t)lvl * plane;
    double *next = nullptr;
    int nw = 0, nh = 0;
    size_t next_stride = 0;
    if (lvl == IMAGES - 3 && oct + 1 < p.octaves) {
        next = p.gauss[oct + 1];
        nw = p.w[oct + 1];
        nh = p.h[oct + 1];
        next_stride = (size_t)IMAGES * nw * nh;
    }
    dim3 grid;
    int strip_h;
    strip_grid<R>(w, h, nframes, grid, strip_h);
    hipLaunchKernelGGL(k_blur_march<R>, grid, dim3(MarchGeom<R>::NTHREADS), MarchGeom<R>::LDS_BYTES, st, in, out, next, w, h, nw, nh,
                       (size_t)IMAGES * plane, next_stride, strip_h, taps);
}

// ---- two consecutive blur levels in one launch ----
//
// The levels whose filters are short enough to leave the vector ALUs idle are bound by the bytes they move: level
// j reads G[j-1] and writes G[j], level j+1 reads G[j] again.  Here one workgroup carries a strip through both
// filters: the rows of G[j] it has just finished go to memory and, by way of an LDS tile, into the row pass and the
// ring of the second filter, so G[j] is never read back.  The first filter works on a tile HB columns wider on
// either side than the strip (the second filter's horizontal support) and RB rows longer at either end of the
// strip.  Operation order per pixel is that of march(), so both levels come out bit-identical to it.
//
// Borders.  Rows of G[j] above and below the image are needed reflected: G[j](-k) = G[j](k).  The first filter
// simply keeps marching over those virtual rows with reflected input rows: its column sum adds the symmetric
// pairs c[R+t] * (v[+t] + v[-t]), reflection swaps the two operands of every pair, and a + b = b + a exactly, so
// the virtual row IS the reflected row.  The row sum runs left to right and has no such symmetry, so columns of
// G[j] left and right of the image are copied from their mirror columns inside the tile before the second filter
// reads them.
constexpr int PAIR_MCH = 16, PAIR_TW = 48;
template <int RA, int RB>
struct PairGeom {
    static constexpr int TW = PAIR_TW;            // output columns of the second level per strip
    static constexpr int HB = 8;                  // halo of the first level's tile on either side
    static constexpr int WA = TW + 2 * HB;        // the first level's tile: whole waves of columns
    static constexpr int MCH = PAIR_MCH;          // rows per step
    static constexpr int CX = 4, CY = 4;          // consecutive outputs per thread in a row pass / a column pass
    static constexpr int NTHREADS = MCH * (WA / CX);
    static constexpr int COLS = WA + 2 * RA;      // input staging tile
    static constexpr int PIN = COLS | 1;
    static constexpr int RINGA = MCH + 2 * RA, PRA = WA + 1;
    static constexpr int PTA = WA + 1;            // tile of first-level rows handed to the second filter
    static constexpr int RINGB = MCH + 2 * RB, PRB = TW + 1;
    // the tile of first-level rows takes the place of the input staging tile, which the first row pass has consumed
    static constexpr size_t LDS_BYTES = ((size_t)MCH * PIN + (size_t)RINGA * PRA + (size_t)RINGB * PRB) * sizeof(double);
    static_assert(PTA <= PIN &&  RB <= HB && WA % 64 == 0 && TW % CX == 0 && MCH % CY == 0 &&
                      (NTHREADS / 64) == (WA / 64) * (MCH / CY) && NTHREADS <= 1024, "geometry");
};

// Row sum of CX consecutive outputs: output m = sum over t = 0..2R, in this order, of c[t] * p[m + t].  The window is
// streamed through registers ECH elements at a time (see march()).
template <int R, int CX>
__device__ __forceinline__ void row_sums(const double *p, const Taps &taps, double (&acc)[CX]) {
    constexpr int NEL = CX + 2 * R, ECH = 4, NCH = (NEL + ECH - 1) / ECH;
    double buf[2][ECH];
#pragma unroll
    for (int u = 0; u < ECH; u++) buf[0][u] = p[u];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int u = 0; u < ECH; u++)
                if ((c + 1) * ECH + u < NEL) buf[(c + 1) & 1][u] = p[(c + 1) * ECH + u];
        }
        stream_fence();
#pragma unroll
        for (int u = 0; u < ECH; u++) {
            const int k = c * ECH + u;
            if (k >= NEL) break;
            const double vk = buf[c & 1][u];
#pragma unroll
            for (int m = 0; m < CX; m++) {
                const int t = k - m;
                if (t == 0) acc[m] = taps.c[0] * vk;
                else if (t > 0 && t <= 2 * R) acc[m] += taps.c[t] * vk;
            }
        }
        stream_fence();
    }
}

// Column sum of CY consecutive output rows at ring column x: output m = c[R] * v[m+R] plus, for t = 1..R in this
// order, c[R+t] * (v[m+R+t] + v[m+R-t]), v[k] = ring row (slot0 + k) mod RING (see march()).
template <int R, int CY, int RING, int PR>
__device__ __forceinline__ void column_sums(const double *ring, int slot0, int x, const Taps &taps, double (&acc)[CY]) {
    auto ld = [&](int k) {
        int sl = slot0 + k;
        sl = sl >= RING ? sl - RING : sl;
        return ring[sl * PR + x];
    };
    constexpr int TCH = 2, NTC = (R + TCH - 1) / TCH;
    double hi_w[CY], lo_w[CY];
    {
        double cen[CY];
#pragma unroll
        for (int m = 0; m < CY; m++) cen[m] = ld(R + m);
#pragma unroll
        for (int m = 0; m < CY; m++) acc[m] = taps.c[R] * cen[m];
#pragma unroll
        for (int m = 0; m + 1 < CY; m++) hi_w[m] = cen[m + 1];
        hi_w[CY - 1] = ld(R + CY);
#pragma unroll
        for (int m = 1; m < CY; m++) lo_w[m] = cen[m - 1];
        lo_w[0] = ld(R - 1);
    }
    double fresh_hi[2][TCH], fresh_lo[2][TCH];
#pragma unroll
    for (int u = 0; u < TCH; u++) {
        const int t = 1 + u;
        if (t < R) {
            fresh_hi[0][u] = ld(R + t + CY);
            fresh_lo[0][u] = ld(R - t - 1);
        }
    }
#pragma unroll
    for (int c = 0; c < NTC; c++) {
        if (c + 1 < NTC) {
#pragma unroll
            for (int u = 0; u < TCH; u++) {
                const int t = 1 + (c + 1) * TCH + u;
                if (t < R) {
                    fresh_hi[(c + 1) & 1][u] = ld(R + t + CY);
                    fresh_lo[(c + 1) & 1][u] = ld(R - t - 1);
                }
            }
        }
        stream_fence();
#pragma unroll
        for (int u = 0; u < TCH; u++) {
            const int t = 1 + c * TCH + u;
            if (t > R) break;
#pragma unroll
            for (int m = 0; m < CY; m++) acc[m] += taps.c[R + t] * (hi_w[m] + lo_w[m]);
            if (t < R) {
#pragma unroll
                for (int m = 0; m + 1 < CY; m++) hi_w[m] = hi_w[m + 1];
                hi_w[CY - 1] = fresh_hi[c & 1][u];
#pragma unroll
                for (int m = CY - 1; m > 0; m--) lo_w[m] = lo_w[m - 1];
                lo_w[0] = fresh_lo[c & 1][u];
            }
        }
#pragma unroll
        for (int m = 0; m < CY; m++) asm volatile("" : "+v"(acc[m]));
        stream_fence();
    }
}

// grid: (strips of TW columns, vertical segments, frames).  in -> outA (filter A) -> outB (filter B).
template <int RA, int RB>
__global__ __launch_bounds__(PAIR_MCH * ((PAIR_TW + 16) / 4)) void k_blur_pair(const double *__restrict__ in, double *__restrict__ outA,
                                                            double *__restrict__ outB, int w, int h, size_t frame_stride,
                                                            int strip_h, Taps tapsA, Taps tapsB) {
    using G = PairGeom<RA, RB>;
    constexpr int NTHREADS = G::NTHREADS, MCH = G::MCH, TW = G::TW, HB = G::HB, WA = G::WA;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *s_in = smem, *tileA = smem, *ringA = s_in + MCH * G::PIN, *ringB = ringA + G::RINGA * G::PRA;
    const TileId tile = xcd_tile_id();
    in += (size_t)tile.z * frame_stride;
    outA += (size_t)tile.z * frame_stride;
    outB += (size_t)tile.z * frame_stride;
    const int tid = threadIdx.x;
    const int tx0 = tile.x * TW, ax0 = tx0 - HB;  // first column of the strip / of the first level's tile
    const int ys = tile.y * strip_h, ye = min(h, ys + strip_h);
    const int nA = ye - ys + 2 * RB;              // first-level rows the strip needs: virtual rows ys - RB ..
    const int nsteps = (ye - ys + 2 * RA + 2 * RB + MCH - 1) / MCH;
    const bool edge_x = ax0 - RA < 0 || ax0 + WA + RA > w;  // uniform over the workgroup

    // input staging: a thread owns one row of the MCH x COLS tile and every TPR-th column of it
    constexpr int TPR = NTHREADS / MCH, NS = (G::COLS + TPR - 1) / TPR;
    const int srow = tid / TPR, sc0 = tid % TPR;
    double pf[NS];
    auto fetch = [&](int s) {
        int gy = ys - RB - RA + MCH * s + srow;
        if ((unsigned)gy >= (unsigned)h) gy = reflect101(gy, h);
        const double *row = in + (size_t)gy * w;
        if (!edge_x) {
            const int gx0 = ax0 - RA + sc0;
#pragma unroll
            for (int u = 0; u < NS; u++)
                if ((u + 1) * TPR <= G::COLS || sc0 + TPR * u < G::COLS) pf[u] = row[gx0 + TPR * u];
        } else {
#pragma unroll
            for (int u = 0; u < NS; u++)
                if ((u + 1) * TPR <= G::COLS || sc0 + TPR * u < G::COLS) pf[u] = row[reflect101(ax0 - RA + sc0 + TPR * u, w)];
        }
    };
    const int row = tid & (MCH - 1), cx = tid / MCH;
    // column passes: a wave is 64 columns of one group of CY rows
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = (wave % (WA / 64)) * 64 + (tid & 63), cy = wave / (WA / 64);
    fetch(0);
    for (int s = 0; s < nsteps; s++) {
        {
            double *q = s_in + srow * G::PIN + sc0;
#pragma unroll
            for (int u = 0; u < NS; u++)
                if ((u + 1) * TPR <= G::COLS || sc0 + TPR * u < G::COLS) q[TPR * u] = pf[u];
        }
        __syncthreads();  // staging tile complete; the previous step is done with every ring and tile
        if (s + 1 < nsteps) fetch(s + 1);
        {   // filter A, row pass: input row (MCH s + row) of the strip into ring A
            double acc[G::CX];
            row_sums<RA, G::CX>(s_in + row * G::PIN + cx * G::CX, tapsA, acc);
            double *q = ringA + ((MCH * s + row) % G::RINGA) * G::PRA + cx * G::CX;
#pragma unroll
            for (int m = 0; m < G::CX; m++) q[m] = acc[m];
        }
        __syncthreads();
        // First-level rows finished by this step: j0 .. j0 + MCH - 1 (row j is virtual row ys - RB + j of that level).
        const int j0 = MCH * s - 2 * RA;
        {   // filter A, column pass: tile row cy*CY + m is first-level row j0 + cy*CY + m
            const int jf = j0 + cy * G::CY;
            if (jf + G::CY > 0 && jf < nA) {
                double acc[G::CY];
                column_sums<RA, G::CY, G::RINGA, G::PRA>(ringA, (jf + 8 * G::RINGA) % G::RINGA, x, tapsA, acc);
                const int gx = ax0 + x;
#pragma unroll
                for (int m = 0; m < G::CY; m++) {
                    tileA[(cy * G::CY + m) * G::PTA + x] = acc[m];
                    const int y = ys - RB + jf + m;
                    if (y >= ys && y < ye && x >= HB && x < HB + TW && gx < w) outA[(size_t)y * w + gx] = acc[m];
                }
            }
        }
        __syncthreads();
        if (edge_x) {  // first-level columns outside the image are their mirror columns
            for (int c = cx; c < WA; c += NTHREADS / MCH) {
                const int gx = ax0 + c;
                if ((unsigned)gx >= (unsigned)w) {
                    const int m = reflect101(gx, w) - ax0;
                    if ((unsigned)m < (unsigned)WA) tileA[row * G::PTA + c] = tileA[row * G::PTA + m];
                }
            }
            __syncthreads();
        }
        {   // filter B, row pass: tile row `row` = first-level row j0 + row into ring B
            const int j = j0 + row;
            if (j >= 0 && j < nA && cx < TW / G::CX) {
                double acc[G::CX];
                row_sums<RB, G::CX>(tileA + row * G::PTA + (HB - RB) + cx * G::CX, tapsB, acc);
                double *q = ringB + (j % G::RINGB) * G::PRB + cx * G::CX;
#pragma unroll
                for (int m = 0; m < G::CX; m++) q[m] = acc[m];
            }
        }
        __syncthreads();
        {   // filter B, column pass: output rows ys + (j0 - 2 RB + cy*CY + m)
            const int b0 = j0 - 2 * RB + cy * G::CY;  // strip-relative output row of m = 0; its window starts at ring row b0
            const int gx = tx0 + x;
            if (b0 + G::CY > 0 && ys + b0 < ye && x < TW && gx < w) {
                double acc[G::CY];
                column_sums<RB, G::CY, G::RINGB, G::PRB>(ringB, (b0 + 8 * G::RINGB) % G::RINGB, x, tapsB, acc);
#pragma unroll
                for (int m = 0; m < G::CY; m++) {
                    const int y = ys + b0 + m;
                    if (y >= ys && y < ye) outB[(size_t)y * w + gx] = acc[m];
                }
            }
        }
    }
}

template <int RA, int RB>
void launch_blur_pair_r(hipStream_t st, const Pyramid &p, int oct, int lvl, int nframes, const Taps &tapsA, const Taps &tapsB) {
    using G = PairGeom<RA, RB>;
    const int w = p.w[oct], h = p.h[oct];
    const size_t plane = (size_t)w * h;
    const int nx = (w + G::TW - 1) / G::TW;
    int ny = (2048 + nx * nframes - 1) / (nx * nframes);
    const int max_ny = (h + 127) / 128;  // a segment repeats 2 (RA + RB) rows of the first filter
    ny = std::max(1, std::min(ny, max_ny));
    int strip_h = ((h + ny - 1) / ny + G::MCH - 1) / G::MCH * G::MCH;
    ny = (h + strip_h - 1) / strip_h;
    hipLaunchKernelGGL((k_blur_pair<RA, RB>), dim3(nx, ny, nframes), dim3(G::NTHREADS), G::LDS_BYTES, st,
                       p.gauss[oct] + (size_t)(lvl - 1) * plane, p.gauss[oct] + (size_t)lvl * plane,
                       p.gauss[oct] + (size_t)(lvl + 1) * plane, w, h, (size_t)IMAGES * plane, strip_h, tapsA, tapsB);
}

// cvtColor(BGR2GRAY) on 8-bit data (sift.cpp:29): OpenCV 4.x integer formula
// with 15-bit coefficients B 3735, G 19235, R 9798 and a rounding shift.
__global__ void k_bgr_to_gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    }
}

// ---- every blur level of the small octaves in one launch ----
//
// An octave of a few thousand pixels does not fill one CU, and its five blur launches cost their launch latency
// five times over; a single frame spends a third of its time on the 15-20 such launches of its upper octaves.
// Here one workgroup per frame holds the octave in LDS and walks through the five levels and on through every
// following octave: level j is the row pass of level j-1 into a scratch plane, then the column pass back (the
// same operation order as march(): taps left to right in the row pass, centre tap first and then the pairs
// outwards in the column pass, borders reflected), each level written to the pyramid as it is finished and the
// even pixels of level 3 kept as the next octave's level 0 (sift.cpp:154-158).
constexpr int SMALL_NT = 1024;

struct SmallTaps {
    Taps lvl[IMAGES];  // [1..IMAGES-1] are used
};

constexpr int SMALL_RMAX = 13;  // widest filter: the reflection tables reach this far past either border

// rx[i] = reflect101(i - SMALL_RMAX, w) and ry likewise for h: the filters index them instead of reflecting.
template <int R>
__device__ __forceinline__ void small_level(double *A, double *T, const int *rx, const int *ry, int w, int h,
                                            const Taps &taps, double *__restrict__ out, double *__restrict__ nxt,
                                            double *N, int nw, int nh) {
    const int n = w * h;
    // Indices first, then every operand, then the sum: the reads of a pixel are all in flight together instead of
    // each waiting for its table entry in turn.
    for (int p = threadIdx.x; p < n; p += SMALL_NT) {
        const int y = p / w, x = p - y * w;
        const double *row = A + y * w;
        const int *cx = rx + x + (SMALL_RMAX - R);  // cx[t] = reflect101(x - R + t, w)
        int at[2 * R + 1];
        double v[2 * R + 1];
#pragma unroll
        for (int t = 0; t <= 2 * R; t++) at[t] = cx[t];
#pragma unroll
        for (int t = 0; t <= 2 * R; t++) v[t] = row[at[t]];
        double acc = taps.c[0] * v[0];
#pragma unroll
        for (int t = 1; t <= 2 * R; t++) acc += taps.c[t] * v[t];
        T[p] = acc;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += SMALL_NT) {
        const int y = p / w, x = p - y * w;
        const int *cy = ry + y + (SMALL_RMAX - R);  // cy[t] = reflect101(y - R + t, h)
        int at[2 * R + 1];
        double v[2 * R + 1];
#pragma unroll
        for (int t = 0; t <= 2 * R; t++) at[t] = cy[t];
#pragma unroll
        for (int t = 0; t <= 2 * R; t++) v[t] = T[at[t] * w + x];
        double acc = taps.c[R] * v[R];
#pragma unroll
        for (int t = 1; t <= R; t++) acc += taps.c[R + t] * (v[R + t] + v[R - t]);
        A[p] = acc;  // the previous level is dead: the row pass has consumed it
        out[p] = acc;
        if (nxt && !(y & 1) && !(x & 1) && (y >> 1) < nh && (x >> 1) < nw) {
            nxt[(y >> 1) * nw + (x >> 1)] = acc;
            N[(y >> 1) * nw + (x >> 1)] = acc;
        }
    }
    __syncthreads();
}

// grid: frames.  LDS: two planes of the first octave taken, one of the octave after it (that octave's level 0)
// and the two reflection tables.
__global__ __launch_bounds__(SMALL_NT) void k_small_octaves(Pyramid p, int o_first, SmallTaps taps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const size_t f = blockIdx.x;
    const int cap = p.w[o_first] * p.h[o_first];
    const int next_cap = o_first + 1 < p.octaves ? p.w[o_first + 1] * p.h[o_first + 1] : 0;
    double *A = smem, *T = smem + cap, *N = smem + 2 * cap;
    int *rx = reinterpret_cast<int *>(N + next_cap);
    int *ry = rx + p.w[o_first] + 2 * SMALL_RMAX;
    for (int o = o_first; o < p.octaves; o++) {
        const int w = p.w[o], h = p.h[o], n = w * h;
        double *G = p.gauss[o] + f * IMAGES * (size_t)n;
        if (o == o_first) {
            for (int i = threadIdx.x; i < n; i += SMALL_NT) A[i] = G[i];
        } else {
            for (int i = threadIdx.x; i < n; i += SMALL_NT) A[i] = N[i];
        }
        for (int i = threadIdx.x; i < w + 2 * SMALL_RMAX; i += SMALL_NT) rx[i] = reflect101(i - SMALL_RMAX, w);
        for (int i = threadIdx.x; i < h + 2 * SMALL_RMAX; i += SMALL_NT) ry[i] = reflect101(i - SMALL_RMAX, h);
        __syncthreads();
        double *nxt = nullptr;
        int nw = 0, nh = 0;
        if (o + 1 < p.octaves) {
            nw = p.w[o + 1];
            nh = p.h[o + 1];
            nxt = p.gauss[o + 1] + f * IMAGES * (size_t)nw * nh;
        }
        static_assert(IMAGES == 6, "five blur levels of fixed radii");
        small_level<5>(A, T, rx, ry, w, h, taps.lvl[1], G + (size_t)1 * n, nullptr, N, 0, 0);
        small_level<6>(A, T, rx, ry, w, h, taps.lvl[2], G + (size_t)2 * n, nullptr, N, 0, 0);
        small_level<8>(A, T, rx, ry, w, h, taps.lvl[3], G + (size_t)3 * n, nxt, N, nw, nh);
        small_level<10>(A, T, rx, ry, w, h, taps.lvl[4], G + (size_t)4 * n, nullptr, N, 0, 0);
        small_level<13>(A, T, rx, ry, w, h, taps.lvl[5], G + (size_t)5 * n, nullptr, N, 0, 0);
    }
}

constexpr size_t SMALL_LDS_MAX = 160 * 1024;

template <int R>
int set_lds_attr() {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_blur_march<R>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)MarchGeom<R>::LDS_BYTES);
    return e1 == hipSuccess ? 0 : -1;
}

}  // namespace

int pyramid_init_attributes() {
    int rc = 0;
    rc |= set_lds_attr<5>();
    rc |= set_lds_attr<6>();
    rc |= set_lds_attr<8>();
    rc |= set_lds_attr<10>();
    rc |= set_lds_attr<13>();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_blur_pair<5, 6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)PairGeom<5, 6>::LDS_BYTES) != hipSuccess)
        rc = -1;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_base_march<5>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)MarchGeom<5>::LDS_BYTES);
    if (e != hipSuccess) rc = -1;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_small_octaves), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)SMALL_LDS_MAX);
    if (e != hipSuccess) rc = -1;
    return rc;
}

void launch_bgr_to_gray(hipStream_t st, const uint8_t *bgr, uint8_t *gray, size_t npix) {
    const int block = 256;
    size_t want = (npix + block - 1) / block;
    const int grid = (int)(want < 2048 ? (want ? want : 1) : 2048);
    hipLaunchKernelGGL(k_bgr_to_gray, dim3(grid), dim3(block), 0, st, bgr, gray, npix);
}

int blur_radius_supported(int r) { return r == 5 || r == 6 || r == 8 || r == 10 || r == 13; }

void launch_base(hipStream_t st, const uint8_t *frames, int W, int H, int nframes, const Pyramid &p, const Taps &taps,
                 int ntaps) {
    (void)ntaps;  // sigma 1.2489996 -> 11 taps
    const int w = 2 * W, h = 2 * H;
    dim3 grid;
    int strip_h;
    strip_grid<5>(w, h, nframes, grid, strip_h);
    hipLaunchKernelGGL(k_base_march<5>, grid, dim3(MarchGeom<5>::NTHREADS), MarchGeom<5>::LDS_BYTES, st, frames, W, H, p.gauss[0],
                       (size_t)IMAGES * w * h, strip_h, taps);
}

void launch_blur(hipStream_t st, const Pyramid &p, int oct, int lvl, int nframes, const Taps &taps, int ntaps) {
    switch (ntaps / 2) {
        case 5: launch_blur_r<5>(st, p, oct, lvl, nframes, taps); break;
        case 6: launch_blur_r<6>(st, p, oct, lvl, nframes, taps); break;
        case 8: launch_blur_r<8>(st, p, oct, lvl, nframes, taps); break;
        case 10: launch_blur_r<10>(st, p, oct, lvl, nframes, taps); break;
        case 13: launch_blur_r<13>(st, p, oct, lvl, nframes, taps); break;
        default: break;  // rejected on the host before any launch
    }
}

bool blur_levels_1_2_pair(const Pyramid &p, int oct, const int *lvl_ntaps) {
    // a narrow octave is mostly halo in 48-column strips, and its launches are short either way
    return lvl_ntaps[1] == 11 && lvl_ntaps[2] == 13 && p.w[oct] >= 4 * PairGeom<5, 6>::TW && p.h[oct] >= 64;
}

void launch_blur_levels_1_2(hipStream_t st, const Pyramid &p, int oct, int nframes, const Taps *lvl_taps) {
    launch_blur_pair_r<5, 6>(st, p, oct, 1, nframes, lvl_taps[1], lvl_taps[2]);
}

int first_small_octave(const Pyramid &p) {
    int o = p.octaves;
    while (o > 0 && p.w[o - 1] * p.h[o - 1] <= SMALL_OCTAVE_PIXELS) o--;
    return o;
}

bool launch_small_octaves(hipStream_t st, const Pyramid &p, int o_first, int nframes, const Taps *lvl_taps,
                          const int *lvl_ntaps) {
    if (o_first >= p.octaves) return true;
    const int radii[IMAGES] = {0, 5, 6, 8, 10, 13};
    SmallTaps taps;
    for (int l = 1; l < IMAGES; l++) {
        if (lvl_ntaps[l] != 2 * radii[l] + 1) return false;
        taps.lvl[l] = lvl_taps[l];
    }
    taps.lvl[0] = lvl_taps[1];  // unused
    const int cap = p.w[o_first] * p.h[o_first];
    const int next = o_first + 1 < p.octaves ? p.w[o_first + 1] * p.h[o_first + 1] : 0;  // octaves only shrink
    const size_t lds = ((size_t)2 * cap + next) * sizeof(double) +
                       (size_t)(p.w[o_first] + p.h[o_first] + 4 * SMALL_RMAX) * sizeof(int);  // planes, reflection tables
    if (cap > SMALL_OCTAVE_PIXELS || lds > SMALL_LDS_MAX) return false;
    hipLaunchKernelGGL(k_small_octaves, dim3(nframes), dim3(SMALL_NT), lds, st, p, o_first, taps);
    return true;
}

}  // namespace sa
