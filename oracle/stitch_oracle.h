/* CPU restatement of the panorama composition (main.cpp:115-145); test infrastructure only.
 * Parity unpinned: see the header of stitch_oracle.c.  Images are 8-bit BGR, tightly packed. */
#ifndef STITCH_ORACLE_H
#define STITCH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::invert of a 3x3 double matrix (closed form); 0 when singular */
int orc_invert3(const double S[9], double D[9]);
/* warpPerspective(src, dst, H, Size(dw, dh)): INTER_LINEAR, BORDER_CONSTANT black; 0 when H is singular */
int orc_warp_perspective(const uint8_t *src, int sw, int sh, const double H[9], int dw, int dh, uint8_t *dst);
/* main.cpp:115-145.  out holds rh rows of up to rw+iw pixels, packed at the cropped width *out_w.
 * 0 when img is taller than ref (the reference throws there) or H is singular. */
int orc_stitch_pair(const uint8_t *ref, int rw, int rh, const uint8_t *img, int iw, int ih, const double H[9],
                    uint8_t *out, int *out_w);

#ifdef __cplusplus
}
#endif
#endif
