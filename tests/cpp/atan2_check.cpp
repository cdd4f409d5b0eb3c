// Host check of sift_amd/csrc/atan2_fast.h against libm: prints the largest
// error in ulps over gradient-like inputs and the count of results that are
// not identical to libm's, and the same for the degrees conversion.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "atan2_fast.h"

static const double kKnots[33][2] = SA_ATAN_KNOTS;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t next() {
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static double uni() { return (double)(next() >> 11) * 0x1p-53; }

static double ulps(double got, double want) {
    if (got == want) return 0;
    int e;
    frexp(want, &e);
    return fabs(got - want) / ldexp(1.0, e - 53);
}

int main() {
    const double PI = 3.14159265358979323846;
    double worst = 0, worst_deg = 0;
    long diff = 0, diff_deg = 0, n = 0;
    auto check = [&](double y, double x) {
        const double want = atan2(y, x), got = sa::atan2_fast(y, x, kKnots);
        const double u = ulps(got, want);
        if (u > worst) worst = u;
        diff += got != want;
        const double wd = (want * 180) / PI, gd = sa::div_pi(got * 180);
        const double ud = ulps(gd, wd);
        if (ud > worst_deg) worst_deg = ud;
        diff_deg += gd != wd;
        // the division by pi alone must be exact
        if (sa::div_pi(want * 180) != wd) { printf("div_pi mismatch at %a\n", want); diff_deg += 1000000; }
        n++;
    };
    // pixel-difference-like values at many scales, all octants
    for (int k = 0; k < 4000000; k++) {
        const double sx = ldexp(1.0, (int)(next() % 40) - 30), sy = ldexp(1.0, (int)(next() % 40) - 30);
        check((uni() * 2 - 1) * sy, (uni() * 2 - 1) * sx);
    }
    // small integers (differences of 8-bit-like data), axes, diagonals, zeros
    for (int a = -40; a <= 40; a++)
        for (int b = -40; b <= 40; b++) check(a * 0.25, b * 0.25);
    // ratios next to the knots and to the interval midpoints
    for (int i = 0; i <= 64; i++)
        for (int k = -200; k <= 200; k++) {
            const double t = i / 64.0 + k * 1e-9;
            check(t, 1.0); check(1.0, t); check(-t, -1.0); check(1.0, -t);
        }
    check(0.0, 0.0); check(0.0, -1.0); check(1e-300, 1e-300); check(3e-310, 1e-305);
    // the orientation-bin quotient (ori*36)/360 (sift.cpp:416) and round(): exact agreement required
    long bin_bad = 0;
    auto check_bin = [&](double ori) {
        const double a = ori * 36, want = a / 360, got = sa::div_360(a);
        if (got != want || sa::round_half_away(got) != (int)round(want)) bin_bad++;
    };
    for (int k = 0; k < 4000000; k++) check_bin((uni() * 2 - 1) * 180.0);
    for (int m = -18; m <= 18; m++)
        for (int k = -2000; k <= 2000; k++) {
            check_bin(nextafter(5.0 + 10.0 * m, 1e9) + k * 1e-13);
            double v = 5.0 + 10.0 * m;
            for (int q = 0; q < (k < 0 ? -k : k) % 40; q++) v = nextafter(v, k < 0 ? -1e9 : 1e9);
            check_bin(v);
        }
    const double halves[] = {0.5, -0.5, 0.49999999999999994, -0.49999999999999994, 1.5, 2.5, -17.5, 17.5, 0.0, -0.0};
    for (double hv : halves)
        if (sa::round_half_away(hv) != (int)round(hv)) bin_bad++;
    printf("bin_mismatches %ld\n", bin_bad);
    printf("samples %ld max_ulp %.3f not_identical %ld max_ulp_deg %.3f not_identical_deg %ld\n", n, worst, diff, worst_deg,
           diff_deg);
    return 0;
}
