// atan2 for gradient directions: one division and a four-term polynomial.
//
// With mx = max(|x|,|y|), mn = min(|x|,|y|) and c = i/32 the multiple of 1/32
// nearest to mn/mx, atan(mn/mx) = atan(c) + atan(u) with
// u = (mn - c*mx) / (mx + c*mn), |u| < 1/63: atan(c) comes from a
// double-double table, atan(u) from u - u^3/3 + u^5/5 - u^7/7 + u^9/9, and the
// octant is undone with double-double pi/2 and pi.  The result is within about
// one ulp of the correctly rounded value (checked against libm on the host by
// tests/test_host.py), which is what the device library's atan2 achieves as
// well, at a third of the instructions.
#pragma once
#include <math.h>

#include "atan_table.h"

#if defined(__HIPCC__)
#define SA_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SA_HD inline
#endif

namespace sa {

// 1/d to about double precision: a hardware estimate refined by two Newton steps.
SA_HD double recip_refined(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
#else
    double r = (double)(1.0f / (float)d);
#endif
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// A double constant held in scalar registers on the device.  A 64-bit literal cannot be an operand, and left to itself
// the compiler moves the addend of every multiply-add of a polynomial into the destination vector registers first
// (two moves per constant and evaluation); an opaque scalar pair makes it take the three-operand form instead.
SA_HD double scalar_const(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    long long bits = __builtin_bit_cast(long long, c);
    asm("" : "+s"(bits));
    return __builtin_bit_cast(double, bits);
#else
    return c;
#endif
}

// `knots` points at the 33 {hi, lo} pairs of SA_ATAN_KNOTS (any address space).
SA_HD double atan2_fast(double y, double x, const double (*knots)[2]) {
    double ax = fabs(x), ay = fabs(y);
    const bool swap = ay > ax;
    double mx = swap ? ay : ax, mn = swap ? ax : ay;
    if (mx < 1e-30) {  // keeps the float estimate below finite; exact scaling
        mx *= 0x1p200;
        mn *= 0x1p200;
    }
    double r = 0.0;
    if (mx > 0.0) {
#if defined(__HIP_DEVICE_COMPILE__)
        const float tf = (float)mn * __builtin_amdgcn_rcpf((float)mx);
#else
        const float tf = (float)mn / (float)mx;
#endif
        int i = (int)rintf(tf * 32.0f);
        i = i < 0 ? 0 : (i > 32 ? 32 : i);
        const double c = (double)i * 0.03125;
        const double num = fma(-c, mx, mn), den = fma(c, mn, mx);
        const double rd = recip_refined(den);
        double u = num * rd;
        u = fma(fma(-u, den, num), rd, u);
        const double s = u * u;
        double p = 1.0 / 9.0;
        p = fma(p, s, scalar_const(-1.0 / 7.0));
        p = fma(p, s, scalar_const(1.0 / 5.0));
        p = fma(p, s, scalar_const(-1.0 / 3.0));
        const double au = fma(u * s, p, u);
        r = knots[i][0] + (knots[i][1] + au);
    }
    if (swap) r = (SA_PIO2_HI - r) + SA_PIO2_LO;
    if (x < 0.0) r = (SA_PI_HI - r) + SA_PI_LO;
    return copysign(r, y);
}

// a / pi, correctly rounded for the magnitudes that occur here (quotient from
// the rounded reciprocal, then one exact-remainder correction).
SA_HD double div_pi(double a) {
    const double q = a * SA_INV_PI;
    return fma(fma(-q, 3.14159265358979323846, a), SA_INV_PI, q);
}

// a / 360 with the IEEE quotient's rounding, for |a| up to a few thousand: same
// reciprocal-multiply-and-correct scheme as div_pi.
SA_HD double div_360(double a) {
    const double inv = 1.0 / 360.0;
    const double q = a * inv;
    return fma(fma(-q, 360.0, a), inv, q);
}

// round() for |v| < 2^31: to nearest, halves away from zero.
SA_HD int round_half_away(double v) {
    double t = trunc(v);
    if (fabs(v - t) >= 0.5) t += copysign(1.0, v);  // v - t is exact
    return (int)t;
}

}  // namespace sa
